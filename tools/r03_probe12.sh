#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_conv_random_gpu.py tests/test_ops_gpu.py tests/test_shapes_gpu.py tests/test_units_gpu.py tests/test_conv_pers_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for tm in 0 1; do MD_W2_TEAMS=$tm timeout -k 10 120 python3 tools/layer_bench.py c1s c1t c3s c3d 2>&1 | grep -v amdgpu | sed -E 's/fwd.*wgrad/wgrad/' | sed "s/^/teams=$tm /"; done
bash tools/r03_ab.sh $1 "MD_W2_TEAMS=0" "MD_W2_TEAMS=1"

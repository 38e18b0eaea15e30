#!/bin/bash
# the tests touched this round + A/B of the BatchNorm-backward reduce geometry (tag = $1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests/test_cfg_fullsize_gpu.py tests/test_vivit.py tests/test_fusion_derived.py tests/test_graphed_gpu.py tests/test_units_gpu.py tests/test_dp_gpu.py tests/test_gb_loops.py tests/test_eval_curve.py -x -q -m gpu -s > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
grep -E "relative L2|passed|failed" $O/tests.log | tail -6
bash tools/r03_ab.sh $1 "MD_X=1" "MD_BN_RED_CAP=2048" 

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_conv_pers_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_train_loop_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
bash tools/r03_ab.sh $1 "MD_FUSE_SLICES=0" "MD_FUSE_SLICES=1"

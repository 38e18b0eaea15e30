#!/bin/bash
# HALF variant of k_conv_patch: conv tests in both settings, unit table, bench A/B (tag = $1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
for h in 2 1; do
MD_PATCH_HALF=$h timeout -k 10 600 python3 -m pytest tests/test_conv_random_gpu.py tests/test_ops_gpu.py tests/test_shapes_gpu.py tests/test_units_gpu.py tests/test_conv_pers_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/tests_h$h.log 2>&1 || { tail -40 $O/tests_h$h.log; exit 1; }
tail -1 $O/tests_h$h.log
done
for h in 0 1 2; do MD_PATCH_HALF=$h timeout -k 10 300 python3 tools/unit_table.py > $O/unit_table_h$h.log 2>&1 || exit 1; tail -1 $O/unit_table_h$h.log; done
bash tools/r03_ab.sh $1 "MD_PATCH_HALF=0" "MD_PATCH_HALF=1" "MD_PATCH_HALF=2"

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for dbg in 0 1 2 4 16 32 48 3; do
  MD_DBG=$dbg timeout -k 10 120 python3 tools/layer_bench.py c1s c1t 2>&1 | grep -v amdgpu | sed -E 's/wgrad.*//' | tee -a $O/abl_c1s.log
done

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for dbg in 0 1 2 4 16 32 3 7; do
  MD_DBG=$dbg timeout -k 10 120 python3 tools/layer_bench.py c3s c3d 2>&1 | grep -v amdgpu | sed -E 's/wgrad.*//' | tee -a $O/abl_patch.log
done

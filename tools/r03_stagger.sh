#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for st in 0 2 4 6 8 12; do
  MD_W2_STAGGER=$st timeout -k 10 120 python3 tools/layer_bench.py c1s c1t c3s 2>&1 | grep -v amdgpu | sed -E 's/fwd.*wgrad/wgrad/' | sed "s/^/stagger=$st /" | tee -a $O/stagger.log
done

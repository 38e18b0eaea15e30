#!/bin/bash
# phase ablation of k_wgrad2 on the two big 64x64 layers (timing only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for dbg in 0 1 2 4 3 5 6 7; do
  MD_DBG2=$dbg timeout -k 10 120 python3 tools/layer_bench.py c1s c1t 2>&1 | grep -v amdgpu | sed "s/^/dbg2=$dbg /" >> $O/abl.log || exit 1
done
for pad in 45 ; do
  MD_W2_PAD_KB=$pad timeout -k 10 120 python3 tools/layer_bench.py c1s c1t 2>&1 | grep -v amdgpu | sed "s/^/pad=$pad /" >> $O/abl.log || exit 1
  MD_W2_PAD_KB=$pad MD_DBG2=2 timeout -k 10 120 python3 tools/layer_bench.py c1s c1t 2>&1 | grep -v amdgpu | sed "s/^/pad=$pad dbg2=2 /" >> $O/abl.log || exit 1
done
for fill in 128 384 512; do
  MD_WGRAD_FILL=$fill timeout -k 10 120 python3 tools/layer_bench.py c1s c1t 2>&1 | grep -v amdgpu | sed "s/^/fill=$fill /" >> $O/abl.log || exit 1
done
cat $O/abl.log | sed -E 's/fwd.*wgrad/wgrad/'

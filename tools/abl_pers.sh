#!/bin/bash
# ablation of the persistent conv kernel phases (MD_DBG bits: 1 no patch loads, 2 no matrix loop, 4 no stores, 8 no commit,
# 16 no issue, 32 no epilogue, 64 no matrix phase body)
mkdir -p gpurun_out
for d in 0 1 2 4 8 16 32 64 24 25 36 60 62 126; do
  MD_DBG=$d timeout -k 10 100 python tools/layer_bench.py c1s c1t 2>/dev/null | grep dbg || exit 1
done

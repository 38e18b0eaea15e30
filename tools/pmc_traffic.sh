#!/bin/bash
# HBM traffic per kernel family (on the GPU box):  bash tools/pmc_traffic.sh <tag>
# Two separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel trace only, as MI355X_MICROARCH.md prescribes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1
mkdir -p $R/gpurun_out/pmc_$T
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$T/$C -o r -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$T/$C.log 2>&1 || exit 1
done

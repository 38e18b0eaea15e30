#!/bin/bash
# BatchNorm-backward reduce pass: grid cap / rows per block / block order, standalone (tools/bn_bench.py) and in the step (bench.py)
set -e
mkdir -p gpurun_out
for cfg in "2048 16 0" "2048 16 1" "4096 16 0" "4096 8 0" "8192 8 0" "1024 32 0" "3072 16 0"; do
  set -- $cfg
  echo "== cap $1 rows $2 fwd $3"
  MD_BN_RED_CAP=$1 MD_BN_RED_ROWS=$2 MD_BN_RED_FWD=$3 python tools/bn_bench.py
  MD_BN_RED_CAP=$1 MD_BN_RED_ROWS=$2 MD_BN_RED_FWD=$3 python bench.py --steps 40 --warmup 10 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
done

#!/bin/bash
# A/B of env configurations on the bench, interleaved twice:  bash tools/r03_ab.sh <tag> "<env1>" "<env2>" ...
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for rep in 1 2; do
  for cfg in "$@"; do
    ( export $cfg; timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])" ) | tee -a $O/ab.log
  done
done

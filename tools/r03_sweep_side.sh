#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
run() { timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a $O/sweep.log; }
MD_WGRAD_STREAM=0 run serial
for r in 0 20000 100000 200000 1000000000; do MD_WGRAD_SIDE_MAX_ROWS=$r run side_rows$r; done
MD_WGRAD_SIDE_MAX_ROWS=100000 MD_WGRAD2_BESIDE=256 run side_rows100000_fill256
MD_WGRAD_SIDE_MAX_ROWS=100000 MD_WGRAD2_BESIDE=96 run side_rows100000_fill96

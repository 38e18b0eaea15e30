#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
run() { timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" | tee -a $O/sweep.log; }
MD_WGRAD_STREAM=0 run serial
for b in 96 128 160 200 256; do MD_WGRAD2_BESIDE=$b run beside$b; done
MD_WGRAD2=0 run firstform
MD_WGRAD2=0 MD_WGRAD_STREAM=0 run firstform_serial

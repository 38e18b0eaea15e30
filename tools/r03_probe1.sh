#!/bin/bash
# round-3 probe: how much of the step is interference between the two streams?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/base.json 2> $O/base.err || exit 1
MD_WGRAD_STREAM=0 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/serial.json 2> $O/serial.err || exit 1
MD_DBG_SKIP_WGRAD=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/nowgrad.json 2> $O/nowgrad.err || exit 1
cd /tmp && export TMPDIR=/tmp
MD_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -o p -- python3 $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline > $O/prof_serial.log 2>&1 || exit 1
cd $R
find $O/prof_serial -name "*kernel_stats.csv" -exec cp {} $O/serial_kernel_stats.csv \;
rm -rf $O/prof_serial

#!/bin/bash
# round-end measurement on the GPU box:  bash tools/final_profile.sh <tag>
# 1. PMC HBM traffic (two passes)  2. rocprofv3 --kernel-trace --stats of bench.py  3. the bench line itself
T=$1; R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_traffic.sh $T || exit 1
# (parse afterwards with: python3 tools/pmc_traffic_parse.py $T 4 ... -- the run has 1 warm-up + 3 steps)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -o p -- python3 $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_$T.log 2>&1 || exit 1
cd $R
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || exit 1
find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -2

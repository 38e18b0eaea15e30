#!/bin/bash
# rocprofv3 kernel summary of the cfg5 step, graphed and eager:  bash tools/cfg5_profile.sh <tag>
T=$1; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CFG5_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5g_$T -o p -- python3 $R/tools/cfg5_smoke.py 4 10 > $R/gpurun_out/prof5g_$T.log 2>&1 || exit 1
NO_CPU_BASELINE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5e_$T -o p -- python3 $R/tools/cfg5_smoke.py 4 10 > $R/gpurun_out/prof5e_$T.log 2>&1 || exit 1
cd $R; find gpurun_out/prof5g_$T gpurun_out/prof5e_$T -name "*kernel_stats.csv"

#!/bin/bash
# rocprofv3 kernel trace of the captured cfg5 step (and, with a second argument, of the eager one):  bash tools/cfg5_profile.sh <tag> [eager]
T=$1; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CFG5_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5g_$T -o p -- python3 $R/tools/cfg5_smoke.py 4 10 > $R/gpurun_out/prof5g_$T.log 2>&1 || exit 1
if [ -n "$2" ]; then
NO_CPU_BASELINE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5e_$T -o p -- python3 $R/tools/cfg5_smoke.py 4 10 > $R/gpurun_out/prof5e_$T.log 2>&1 || exit 1
fi
cd $R; python3 tools/trace_streams.py gpurun_out/prof5g_$T/p_kernel_trace.csv > gpurun_out/prof5g_$T.streams.txt; cat gpurun_out/prof5g_$T.streams.txt

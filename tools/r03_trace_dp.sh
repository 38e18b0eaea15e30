#!/bin/bash
# rocprofv3 kernel stats of the forced one-rank data-parallel bench leg: bash tools/r03_trace_dp.sh <tag>
R=$GRAFT_REPO_ROOT; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MD_BENCH_FORCE_DP=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o p -- python3 $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline > $O/tr.log 2>&1 || { tail -20 $O/tr.log; exit 1; }
f=$(find $O/tr -name "*kernel_stats.csv" | head -1)
cp $f $O/kstats.csv
python3 $R/tools/kstats.py $f "" 60
rm -rf $O/tr

#!/bin/bash
# rocprofv3 kernel stats of bench.py under a few env configurations: bash tools/r03_trace_bench.sh <tag> "<env1>" "<env2>" ...
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg true
  ( export $cfg; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$i -o p -- python3 $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline > $O/tr$i.log 2>&1 ) || exit 1
  f=$(find $O/tr$i -name "*kernel_stats.csv" | head -1)
  cp $f $O/kstats_$i.csv
  echo "== cfg $i: $cfg"; python3 $R/tools/kstats.py $f "" 14
  rm -rf $O/tr$i
done

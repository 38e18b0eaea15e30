#!/bin/bash
# rocprofv3 kernel stats of the ViViT cfg3 captured step: bash tools/r03_trace_vivit.sh <tag>
R=$GRAFT_REPO_ROOT; T=$1; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o p -- python3 $R/tools/vivit_graph.py 40 > $O/tr.log 2>&1 || { tail -20 $O/tr.log; exit 1; }
f=$(find $O/tr -name "*kernel_stats.csv" | head -1)
cp $f $O/kstats.csv
python3 $R/tools/kstats.py $f "" 45
rm -rf $O/tr

#!/bin/bash
# kernel-level durations of the second-form weight gradient on single layers (rocprofv3 kernel trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dbg in 0 7 5; do
  MD_DBG2=$dbg timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$dbg -o p -- python3 $R/tools/layer_bench.py c1s c1t c3s c3d > $O/tr$dbg.log 2>&1 || exit 1
  f=$(find $O/tr$dbg -name "*kernel_stats.csv" | head -1)
  echo "== dbg2=$dbg"; python3 $R/tools/kstats.py $f k_wgrad2 | tee $O/w2_dbg$dbg.txt
  rm -rf $O/tr$dbg
done

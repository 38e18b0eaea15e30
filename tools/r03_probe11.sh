#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_conv_pers_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for i in 1 2; do timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o p -- python3 $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline > $O/tr.log 2>&1 || exit 1
f=$(find $O/tr -name "*kernel_stats.csv" | head -1); cp $f $O/kstats.csv; rm -rf $O/tr
python3 $R/tools/kstats.py $O/kstats.csv k_conv_pers

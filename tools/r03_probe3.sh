#!/bin/bash
# wgrad-focused probe: conv tests, layer ablation (c1s c1t), unit table, bench both schedules (tag = $1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests/test_conv_random_gpu.py tests/test_ops_gpu.py tests/test_shapes_gpu.py tests/test_units_gpu.py tests/test_conv_pers_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for dbg in 0 7 5; do
  MD_DBG2=$dbg timeout -k 10 120 python3 tools/layer_bench.py c1s c1t c3s c3d 2>&1 | grep -v amdgpu | sed "s/^/dbg2=$dbg /" >> $O/abl.log || exit 1
done
sed -E 's/fwd.*wgrad/wgrad/' $O/abl.log
timeout -k 10 300 python3 tools/unit_table.py > $O/unit_table.log 2>&1 || exit 1
tail -1 $O/unit_table.log
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/base.json 2> $O/base.err || exit 1
MD_WGRAD_STREAM=0 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/serial.json 2> $O/serial.err || exit 1
python3 -c "
import json
for f in ('base','serial'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'])
"

#!/bin/bash
# round-3 probe: correctness of a kernel change on the conv tests, then per-unit times and the bench line (tag = $1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests/test_conv_random_gpu.py tests/test_ops_gpu.py tests/test_shapes_gpu.py tests/test_units_gpu.py tests/test_conv_pers_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python3 tools/unit_table.py > $O/unit_table.log 2>&1 || exit 1
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/base.json 2> $O/base.err || exit 1
MD_WGRAD_STREAM=0 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/serial.json 2> $O/serial.err || exit 1
python3 -c "
import json
for f in ('base','serial'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'])
"

#!/bin/bash
# ViViT cfg3 after a change to its launch sequence: tests, then the captured step (A/B through the env switch given as $1), then a trace
set -e
mkdir -p gpurun_out
python -m pytest tests/test_vivit.py tests/test_conv_pers_gpu.py tests/test_ops_gpu.py tests/test_conv_random_gpu.py tests/test_slowfast.py tests/test_graphed_gpu.py -m gpu -x -q > gpurun_out/vivit_tests.log 2>&1 || { tail -30 gpurun_out/vivit_tests.log; exit 1; }
tail -3 gpurun_out/vivit_tests.log
for i in 1 2; do
  env $1 python3 tools/vivit_graph.py 100 > gpurun_out/vivit_a_$i.json
  python3 tools/vivit_graph.py 100 > gpurun_out/vivit_b_$i.json
done
cat gpurun_out/vivit_a_*.json gpurun_out/vivit_b_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['value'], j['ms_per_step'], j['loss'])"
bash tools/r03_trace_vivit.sh vv2 > gpurun_out/vv2.txt 2>&1

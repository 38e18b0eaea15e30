#!/bin/bash
# ViViT cfg3 captured step, A/B through the env switch given as $1 (A = with the switch), then a kernel trace of B
set -e
mkdir -p gpurun_out
for i in 1 2; do
  env $1 python3 tools/vivit_graph.py 100 > gpurun_out/vivit_a_$i.json
  python3 tools/vivit_graph.py 100 > gpurun_out/vivit_b_$i.json
done
cat gpurun_out/vivit_a_*.json gpurun_out/vivit_b_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['value'], j['ms_per_step'], j['loss'])"
bash tools/r03_trace_vivit.sh $2 > gpurun_out/$2.txt 2>&1

#!/bin/bash
# ViViT cfg3 with the Linear weight gradients through the second form (rows walked as an H' x W' grid; channel slices) vs the first form
set -e
mkdir -p gpurun_out
python -m pytest tests/test_vivit.py -m gpu -x -q > gpurun_out/vivit_tests.log 2>&1 || { tail -30 gpurun_out/vivit_tests.log; exit 1; }
tail -3 gpurun_out/vivit_tests.log
python -m pytest tests/test_conv_pers_gpu.py tests/test_ops_gpu.py -m gpu -x -q > gpurun_out/conv_tests.log 2>&1 || { tail -30 gpurun_out/conv_tests.log; exit 1; }
tail -3 gpurun_out/conv_tests.log
for i in 1 2; do
  MD_WGRAD2=0 python3 tools/vivit_graph.py 100 > gpurun_out/vivit_w2_off_$i.json
  python3 tools/vivit_graph.py 100 > gpurun_out/vivit_w2_on_$i.json
done
cat gpurun_out/vivit_w2_off_*.json gpurun_out/vivit_w2_on_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['value'], j['ms_per_step'], j['loss'])"

#!/bin/bash
# HIP runtime settings against the step times (R(2+1)D bench eager; cfg5 and cfg3 captured):  bash tools/runtime_knobs.sh
b() { python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('  bench', d['value'], d['ms_per_step'])"; }
g5() { CFG5_GRAPH=1 python tools/cfg5_smoke.py 4 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('  cfg5 graph', d['ms_per_step'])"; }
g3() { python tools/vivit_graph.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('  cfg3 graph', d['ms_per_step'])"; }
echo "== defaults"; b; g5; g3
for kv in DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 HSA_ENABLE_INTERRUPT=0 DEBUG_HIP_GRAPH_DOT_PRINT=0 AMD_SERIALIZE_KERNEL=0 HIP_FORCE_DEV_KERNARG=0; do
  echo "== $kv"; env $kv bash -c "$(declare -f b g5 g3); b; g5; g3"
done

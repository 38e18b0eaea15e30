#!/bin/bash
# captured-step time of cfg5 (SlowFast + MLSTM_FCN) and cfg3 (ViViT) against the library's sizing knobs:  bash tools/knob_sweep.sh
run5() { CFG5_GRAPH=1 python tools/cfg5_smoke.py 4 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg5', d['ms_per_step'])"; }
run3() { python tools/vivit_graph.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('cfg3', d['ms_per_step'])"; }
echo "== defaults"; run5; run5; run3
for v in 64 256 512; do echo "== MD_PATCH_FILL=$v"; MD_PATCH_FILL=$v run5; done
for v in 128 256; do echo "== MD_GEMM_FILL=$v"; MD_GEMM_FILL=$v run5; done
for v in 0 2; do echo "== MD_PATCH_W8=$v"; MD_PATCH_W8=$v run5; MD_PATCH_W8=$v run3; done
echo "== MD_WGRAD_W8=0"; MD_WGRAD_W8=0 run5; MD_WGRAD_W8=0 run3
echo "== MD_WGRAD_PF=0"; MD_WGRAD_PF=0 run5; MD_WGRAD_PF=0 run3
echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 run5
echo "== MD_PERS=0"; MD_PERS=0 run5

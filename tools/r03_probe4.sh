#!/bin/bash
# full GPU test suite subset relevant to the R(2+1)D path + bench A/B of this round's switches (tag = $1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_conv_random_gpu.py tests/test_ops_gpu.py tests/test_shapes_gpu.py tests/test_units_gpu.py tests/test_conv_pers_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_train_loop_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() { timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('step_trace',{}).get('host_queue_ms'))" | tee -a $O/sweep.log; }
run default
MD_WGRAD_BATCH_REDUCE=0 run nobatch
MD_BN_FUSED_FIN=0 run nofusedfin
MD_WGRAD_BATCH_REDUCE=0 MD_BN_FUSED_FIN=0 run neither
MD_WGRAD_STREAM=1 run sidestream
run default2

#!/bin/bash
# step time against the number of CUs the side-stream weight-gradient kernels are sized for (MD_WGRAD_FILL)
for f in 64 96 128 160 192 256; do
  echo -n "MD_WGRAD_FILL=$f "
  MD_WGRAD_FILL=$f timeout -k 10 200 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done

"""Which ATen operators (not this library's kernels) launch GPU work in one cfg5 step, and from where?   python tools/r03_cfg5_ops.py"""
import os, sys
os.environ["NO_CPU_BASELINE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0], "4", "3"]
src = open(os.path.join(ROOT, "tools", "cfg5_smoke.py")).read()
src = src[:src.index("for _ in range(3):\n    step()")]          # model, data and step() only
exec(compile(src, "cfg5_smoke_head", "exec"))
import torch
from torch.profiler import profile, ProfilerActivity
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=4) if e.key.startswith("aten::") and (getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)) > 0]
rows.sort(key=lambda e: -e.count)
seen = 0
for e in rows[:40]:
    dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    st = [s for s in (e.stack or []) if "disruption" in s or "tools/" in s][:2]
    print(f"{e.key:34s} x{e.count:4d}  {dt:8.1f} us   {' <- '.join(s.split('/')[-1][:60] for s in st)}")

"""Per-unit timing of the three convolution kernels for every unit of the BASELINE trunk (run on the GPU box):
    python tools/unit_table.py
Columns: time in us for forward / data gradient / weight gradient, and the HBM floor (fp32 in + out at 8 TB/s)."""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
from src import ops
from src._plan import TrunkPlan

plan = TrunkPlan(8, 21, 128, 128, [1, 2, 2, 1], 0.01)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
seen = {}
tot = [0.0, 0.0, 0.0, 0.0]
for i, d in enumerate(plan.descs):
    key = tuple(getattr(d, f) for f, _ in d._fields_)
    if key not in seen:
        x = torch.randn(d.N, d.Ti, d.Hi, d.Wi, ops.cpad(d.Cin), device='cuda')
        w = torch.randn(d.Cout, d.Cin, d.kt, d.kh, d.kw, device='cuda') * 0.05
        sc = torch.rand(ops.cpad(d.Cin), device='cuda') + 0.5; sh = torch.randn(ops.cpad(d.Cin), device='cuda') * 0.1
        wf, wd = ops.pack_weights(d, w)
        dy = torch.randn(d.N, d.To, d.Ho, d.Wo, ops.cpad(d.Cout), device='cuda')
        v = ops.view(x, sc, sh, 0.01)
        tf = timeit(lambda: ops.conv_fwd(d, v, wf, 'cuda:0', True))
        td = timeit(lambda: ops.conv_dgrad(d, dy, wd)) if i else 0.0
        tw = timeit(lambda: ops.conv_wgrad(d, v, dy))
        fl = 2.0 * d.N * d.To * d.Ho * d.Wo * d.Cout * d.Cin * d.kt * d.kh * d.kw
        io = 4.0 * (x.numel() + dy.numel()) / 8e6
        seen[key] = (tf, td, tw, io, fl)
        del x, w, dy
    tf, td, tw, io, fl = seen[key]
    tot[0] += tf; tot[1] += td; tot[2] += tw; tot[3] += io
    print(f"u{i:02d} {d.Cin:3d}->{d.Cout:3d} k{d.kt}{d.kh}{d.kw} s{d.st}{d.sh}{d.sw} in {d.Ti}x{d.Hi}x{d.Wi}: fwd {tf:6.1f} dgrad {td:6.1f} wgrad {tw:6.1f} | floor {io:5.1f} us  {fl/1e9:6.1f} GF")
print(f"total fwd {tot[0]:.0f} dgrad {tot[1]:.0f} wgrad {tot[2]:.0f} floor/op {tot[3]:.0f} us")

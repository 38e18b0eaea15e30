"""ViViT cfg3 FeedForward Linears in isolation (rows 16548): forward / data gradient / weight gradient times, with the per-box kernels'
MD_DBG ablation bits (1 no patch loads, 2 no matrix loop, 4 no stores).   MD_DBG=<bits> python tools/r03_linear_abl.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
from src import ops
rows = int(os.environ.get("ROWS", "16548"))
for name, din, dout in (("ff1 128->1024", 128, 1024), ("ff2 1024->128", 1024, 128), ("qkv 128->384", 128, 384), ("out 256->128", 256, 128)):
    d = ops.make_desc(1, 1, 1, rows, din, dout, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    x = torch.randn(rows, din, device="cuda"); w = torch.randn(dout, din, device="cuda") * 0.05; dy = torch.randn(rows, dout, device="cuda")
    wf, wd = ops.pack_weights(d, w[:, :, None, None, None].contiguous())
    def timeit(f, n=30):
        for _ in range(3): f()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
    tf = timeit(lambda: ops.conv_fwd(d, ops.view(x), wf, x.device, want_stats=False))
    td = timeit(lambda: ops.conv_dgrad(d, dy, wd))
    tw = timeit(lambda: ops.conv_wgrad(d, ops.view(x), dy))
    io = 4.0 * rows * (din + dout)
    print(f"{name} dbg={os.environ.get('MD_DBG', '0')}: fwd {tf:6.1f} us  dgrad {td:6.1f} us  wgrad {tw:6.1f} us | io floor {io / 5e6:5.1f} us, mfma floor {2.0 * rows * din * dout * 3 / 1.7e9:5.1f} us")

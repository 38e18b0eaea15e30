"""Streaming rate of the BatchNorm-backward passes at BASELINE layer shapes (run on the GPU box):  python tools/bn_bench.py"""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import ctypes as C, torch
from src import ops, _native as N
L = N.lib()
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, rows, Cc in [('c1 out 64ch 64x64', 8*21*64*64, 64), ('c1s out 72ch 64x64', 8*21*64*64, 72), ('c1t out 32ch 64x64', 8*21*64*64, 32), ('c3s out 144ch 32x32', 8*11*32*32, 144), ('stem mid 45ch', 8*21*64*64, 45)]:
    Cp = ops.cpad(Cc)
    y = torch.randn(rows, Cp, device='cuda'); dA = torch.randn(rows, Cp, device='cuda')
    st = torch.rand(4, Cp, device='cuda') + 0.5
    main = ops.view(y, st[2], st[3], 0.01)
    nb = L.md_bn_bwd_blocks(rows, Cc)
    part = torch.empty(nb, 2, Cp, device='cuda'); coef = torch.rand(2, Cp, device='cuda') * 0.01
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tr = timeit(lambda: L.md_bn_bwd_reduce(ops._p(dA), C.byref(main), None, 1.0, ops._p(st[0]), ops._p(st[1]), rows, Cc, ops._p(part), s))
    ta = timeit(lambda: L.md_bn_bwd_apply(ops._p(dA), C.byref(main), None, 1.0, ops._p(st[0]), ops._p(st[1]), ops._p(coef), rows, Cc, ops._p(dA), None, s))
    by = rows * Cp * 4
    print(f"{name:22s} blocks {nb:5d}: reduce {tr:6.1f} us = {2*by/tr/1e6:5.2f} TB/s read | apply {ta:6.1f} us = {3*by/ta/1e6:5.2f} TB/s (2 read + 1 write)")

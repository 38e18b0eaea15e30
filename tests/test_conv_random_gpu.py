"""GPU: randomised geometries (ragged sizes, boxes cut by the tensor edge, odd channel counts, strides 1-3, kernels up
to 5, asymmetric padding per axis, batch 1) for conv forward / data gradient / weight gradient in both arithmetic modes
against F.conv3d on the CPU.  Exercises the patch kernels (unit stride and power-of-two strided), the strided
data-gradient form and the exact-fp32 gather kernels (stride 3 and anything the patch kernels decline)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src import ops

from tests.test_ops_gpu import cl, uncl, relerr

DEV = "cuda:0"


def random_case(rng):
    while True:
        k = tuple(int(rng.choice([1, 1, 3, 3, 5])) if rng.random() < 0.7 else int(rng.integers(1, 4)) for _ in range(3))
        s = tuple(int(rng.choice([1, 1, 1, 2, 2, 3])) for _ in range(3))
        p = tuple(int(rng.integers(0, kk // 2 + 1)) for kk in k)
        dims = (int(rng.integers(1, 4)), int(rng.integers(1, 9)), int(rng.integers(3, 21)), int(rng.integers(3, 21)))
        cin = int(rng.choice([1, 3, 5, 8, 21, 32, 45, 72]))
        cout = int(rng.choice([2, 7, 16, 33, 64, 115, 150]))
        ok = all((d + 2 * pp - kk) // ss + 1 >= 1 and d + 2 * pp >= kk for d, pp, kk, ss in zip(dims[1:], p, k, s))
        if ok and cin * k[0] * k[1] * k[2] * cout < 400000:
            return cin, cout, k, s, p, dims


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_geometry(seed):
    rng = np.random.default_rng(1000 + seed)
    cin, cout, k, s, p, (N, T, H, W) = random_case(rng)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, cin, T, H, W, generator=g)
    w = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    y = F.conv3d(xr, wr, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    d = ops.make_desc(N, T, H, W, cin, cout, k, s, p)
    xg, dyg, wg = cl(x).to(DEV), cl(dy).to(DEV), w.to(DEV)
    for exact in (False, True):
        ops.set_exact_fp32(exact)
        try:
            wf, wd = ops.pack_weights(d, wg)
            yg, part = ops.conv_fwd(d, ops.view(xg), wf, DEV, want_stats=True)
            dx = ops.conv_dgrad(d, dyg, wd)
            dw = ops.conv_wgrad(d, ops.view(xg), dyg)
            torch.cuda.synchronize()
        finally:
            ops.set_exact_fp32(False)
        tag = (seed, exact, cin, cout, k, s, p, (N, T, H, W))
        assert relerr(uncl(yg.cpu(), cout), y.detach()) < 3e-5, tag
        assert relerr(uncl(dx.cpu(), cin), xr.grad) < 5e-5, tag
        assert relerr(dw.cpu(), wr.grad) < 5e-5, tag
        s1 = part[:, 0, :cout].double().sum(0).cpu()
        ref = y.detach().double().sum(dim=(0, 2, 3, 4))
        assert float((s1 - ref).abs().max()) < 1e-3 * max(1.0, float(ref.abs().max())), tag
        if yg.shape[-1] > cout:
            assert float(yg[..., cout:].abs().max()) == 0.0, tag

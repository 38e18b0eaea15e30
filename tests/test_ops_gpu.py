"""GPU parity: every C-ABI op (through the ctypes binding) against plain fp32 PyTorch-CPU / the oracle.

Tolerance: 2e-5 of the tensor's max magnitude for forward values and gradients (exact-fp32 MFMA with a
different summation order than ATen's CPU kernels), stated per assert.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src import ops  # noqa: E402  (package dir is put on sys.path by conftest)

DEV = "cuda:0"


def cl(x):
    """CPU (B,C,T,H,W) -> channels-last padded [B,T,H,W,Cp]"""
    B, C, T, H, W = x.shape
    Cp = (C + 3) & ~3
    out = torch.zeros(B, T, H, W, Cp)
    out[..., :C] = x.permute(0, 2, 3, 4, 1)
    return out


def uncl(x, C):
    return x[..., :C].permute(0, 4, 1, 2, 3).contiguous()


def relerr(a, b):
    a = a.double(); b = b.double()
    return float((a - b).abs().max() / max(1e-12, float(b.abs().max())))


CASES = [
    # name, Cin, Cout, kernel, stride, pad, (N,T,H,W)
    ("sp3x3", 32, 72, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 12, 10)),
    ("stem7x7s2", 3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3), (2, 2, 22, 18)),
    ("tmp3", 72, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 5, 6, 7)),
    ("sp3x3s2", 32, 115, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 2, 13, 12)),
    ("tmp3s2", 115, 64, (3, 1, 1), (2, 1, 1), (1, 0, 0), (2, 7, 5, 5)),
    ("skip1x1s2", 32, 21, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 3, 9, 8)),
    ("skipt1s2", 21, 64, (1, 1, 1), (2, 1, 1), (0, 0, 0), (2, 5, 4, 4)),
    ("wide230", 64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 2, 8, 8)),
    ("wide288", 128, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 6, 6)),
    ("tmp288", 288, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 3, 4, 4)),
    ("full3d", 5, 9, (3, 3, 3), (2, 1, 2), (1, 1, 0), (2, 6, 7, 9)),   # general 3-D conv (SlowFast laterals use kt>1 strided)
    # <= 4 input channels, W-stride 2, even width: the pixel-pair form of the patch kernels (every kw / pad parity)
    ("pair3x3p1", 3, 20, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 2, 10, 12)),
    ("pair2x2p0", 4, 16, (1, 2, 2), (1, 2, 2), (0, 0, 0), (1, 3, 8, 14)),
    ("pair5p2", 1, 33, (3, 5, 5), (1, 2, 2), (1, 2, 2), (2, 3, 9, 16)),
    ("pair4p1w", 2, 7, (1, 3, 4), (1, 1, 2), (0, 1, 1), (2, 2, 7, 10)),
    ("pair2p2", 3, 5, (1, 1, 2), (1, 1, 2), (0, 0, 2), (1, 2, 5, 8)),
    ("pair7p0", 3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 0), (1, 2, 20, 134)),
    # strided data gradient by residue classes (used from 65536 destination pixels up): odd sizes, every stride pattern
    ("cls_s122", 16, 24, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 2, 131, 128)),
    ("cls_s211", 24, 16, (3, 1, 1), (2, 1, 1), (1, 0, 0), (2, 9, 64, 64)),
    ("cls_s222", 8, 12, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 9, 90, 93)),
    ("cls_k2p0", 8, 8, (1, 2, 2), (1, 2, 2), (0, 0, 0), (1, 2, 182, 182)),
    ("cls_k5p2", 4, 20, (1, 5, 4), (1, 2, 2), (0, 2, 1), (1, 3, 150, 151)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("prologue", [False, True])
def test_conv_fwd_dgrad_wgrad(case, prologue):
    name, Cin, Cout, k, s, p, (Nn, T, H, W) = case
    g = torch.Generator().manual_seed(hash(name) % 1000)
    x = torch.randn(Nn, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / np.sqrt(Cin * k[0] * k[1] * k[2])
    slope = 0.1
    if prologue:
        sc = torch.rand(Cin, generator=g) + 0.5
        sh = torch.randn(Cin, generator=g) * 0.3
        a = F.leaky_relu(x * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1), slope)
    else:
        a = x
    a = a.detach().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.conv3d(a, wr, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)

    d = ops.make_desc(Nn, T, H, W, Cin, Cout, k, s, p)
    xg = cl(x).to(DEV)
    Cpi = xg.shape[-1]
    if prologue:
        scp = torch.zeros(Cpi); scp[:Cin] = sc
        shp = torch.zeros(Cpi); shp[:Cin] = sh
        scg, shg = scp.to(DEV), shp.to(DEV)
        v = ops.view(xg, scg, shg, slope)
    else:
        v = ops.view(xg)
    wg = w.to(DEV)
    wf, wd = ops.pack_weights(d, wg)
    yg, part = ops.conv_fwd(d, v, wf, DEV, want_stats=True)
    torch.cuda.synchronize()
    y_hip = uncl(yg.cpu(), Cout)
    assert relerr(y_hip, y.detach()) < 3e-5
    # pad channels stay exactly zero
    assert float(yg[..., Cout:].abs().max().cpu()) == 0.0 if yg.shape[-1] > Cout else True
    # BN partial statistics from the epilogue
    s1 = part[:, 0, :Cout].double().sum(0).cpu(); s2 = part[:, 1, :Cout].double().sum(0).cpu()
    yd = y.detach().double()
    assert relerr(s1, yd.sum(dim=(0, 2, 3, 4))) < 1e-4 or float((s1 - yd.sum(dim=(0, 2, 3, 4))).abs().max()) < 1e-3
    assert relerr(s2, (yd * yd).sum(dim=(0, 2, 3, 4))) < 2e-5

    dyg = cl(dy).to(DEV)
    dx = ops.conv_dgrad(d, dyg, wd)
    dw = ops.conv_wgrad(d, v, dyg)
    torch.cuda.synchronize()
    assert relerr(uncl(dx.cpu(), Cin), a.grad) < 3e-5
    assert relerr(dw.cpu(), wr.grad) < 2e-5
    # accumulate form: dx2 = dx + dgrad
    dx2 = ops.conv_dgrad(d, dyg, wd, out=dx.clone(), accumulate=True)
    torch.cuda.synchronize()
    assert relerr(dx2.cpu(), 2 * dx.cpu()) < 1e-6


def test_conv_rejects_cpu_and_bad_shapes():
    d = ops.make_desc(1, 2, 8, 8, 4, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    with pytest.raises(RuntimeError):
        ops.view(torch.zeros(1, 2, 8, 8, 4))           # CPU tensor
    bad = ops.make_desc(1, 2, 8, 8, 4, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    bad.To = 5
    with pytest.raises(RuntimeError):
        ops.pack_weights(bad, torch.zeros(8, 4, 1, 3, 3, device=DEV))


@pytest.mark.parametrize("C,rows_shape", [(45, (2, 3, 10, 9)), (72, (2, 2, 16, 16)), (288, (1, 2, 5, 5)), (21, (3, 1, 7, 7))])
def test_bn_forward_backward_unit(C, rows_shape):
    g = torch.Generator().manual_seed(C)
    Nn, T, H, W = rows_shape
    raw = torch.randn(Nn, C, T, H, W, generator=g) * 2 + 0.7
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    rm = torch.zeros(C); rv = torch.ones(C)
    slope = 0.01
    rawr = raw.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    a = F.leaky_relu(F.batch_norm(rawr, rm, rv, gr, br, True, 0.1, 1e-5), slope)
    dA = torch.randn(a.shape, generator=g)
    a.backward(dA)

    rawg = cl(raw).to(DEV)
    rows = Nn * T * H * W
    Cp = rawg.shape[-1]
    # partials as the conv epilogue would produce them (two fake row blocks)
    flat = rawg.view(rows, Cp)
    half = rows // 2
    part = torch.stack([torch.stack([flat[:half].sum(0), (flat[:half] ** 2).sum(0)]),
                        torch.stack([flat[half:].sum(0), (flat[half:] ** 2).sum(0)])]).contiguous()
    rmg, rvg = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    st = ops.bn_finalize(part, C, rows, gamma.to(DEV), beta.to(DEV), rmg, rvg)
    v = ops.view(rawg, st[2], st[3], slope)
    ag = ops.bn_act(v, rawg, C)
    torch.cuda.synchronize()
    assert relerr(uncl(ag.cpu(), C), a.detach()) < 2e-5
    assert relerr(rmg.cpu(), rm) < 1e-5 and relerr(rvg.cpu(), rv) < 1e-5      # running stats (rm/rv were updated in place by F.batch_norm)
    d_raw, dS, dgam, dbet = ops.bn_backward(cl(dA).to(DEV), v, st, C)
    torch.cuda.synchronize()
    assert dS is None
    assert relerr(uncl(d_raw.cpu(), C), rawr.grad) < 5e-5
    assert relerr(dgam.cpu(), gr.grad) < 2e-5 and relerr(dbet.cpu(), br.grad) < 2e-5


@pytest.mark.parametrize("C,rows_shape", [(72, (2, 8, 64, 64)), (32, (4, 8, 64, 64)), (144, (1, 2, 12, 12))])
def test_bn_backward_fused_finalize_large_and_small(C, rows_shape):
    """BatchNorm-backward with the finalize folded into the apply pass (md_bn_bwd_apply_fused) against the three-launch form and against autograd in fp64."""
    g = torch.Generator().manual_seed(C + 1)
    Nn, T, H, W = rows_shape
    raw = torch.randn(Nn, C, T, H, W, generator=g) * 1.5 + 0.3
    gamma = torch.rand(C, generator=g) + 0.5; beta = torch.randn(C, generator=g) * 0.2
    slope = 0.01
    rawr = raw.double().requires_grad_(True); gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
    a = F.leaky_relu(F.batch_norm(rawr, None, None, gr, br, True, 0.1, 1e-5), slope)
    dA = torch.randn(a.shape, generator=g)
    a.backward(dA.double())
    rawg = cl(raw).to(DEV)
    rows = Nn * T * H * W
    flat = rawg.view(rows, -1).double()
    part = torch.stack([flat.sum(0), (flat ** 2).sum(0)]).unsqueeze(0).float().contiguous()
    st = ops.bn_finalize(part, C, rows, gamma.to(DEV), beta.to(DEV))
    v = ops.view(rawg, st[2], st[3], slope)
    dAg = cl(dA).to(DEV)
    fused = ops.bn_backward(dAg, v, st, C, fused_finalize=True)
    plain = ops.bn_backward(dAg, v, st, C, fused_finalize=False)
    torch.cuda.synchronize()
    for f, p_ in zip((fused[0], fused[2], fused[3]), (plain[0], plain[2], plain[3])):
        assert relerr(f.cpu(), p_.cpu()) < 1e-6
    assert relerr(uncl(fused[0].cpu(), C), rawr.grad.float()) < 5e-5
    assert relerr(fused[2].cpu(), gr.grad.float()) < 2e-5 and relerr(fused[3].cpu(), br.grad.float()) < 2e-5


@pytest.mark.parametrize("skip_is_view", [False, True])
def test_residual_close_forward_backward(skip_is_view):
    g = torch.Generator().manual_seed(5)
    Nn, C, T, H, W = 2, 64, 3, 6, 5
    alpha, slope = 0.2, 0.01
    raw = torch.randn(Nn, C, T, H, W, generator=g)
    xs = torch.randn(Nn, C, T, H, W, generator=g)
    gamma = torch.rand(C, generator=g) + 0.5; beta = torch.randn(C, generator=g) * 0.2
    gs = torch.rand(C, generator=g) + 0.5; bs = torch.randn(C, generator=g) * 0.2
    rawr = raw.clone().requires_grad_(True); xr = xs.clone().requires_grad_(True)
    main = F.leaky_relu(F.batch_norm(rawr, None, None, gamma, beta, True, 0.1, 1e-5), slope)
    if skip_is_view:
        sk = F.leaky_relu(xr * gs.view(1, -1, 1, 1, 1) + bs.view(1, -1, 1, 1, 1), slope)
    else:
        sk = xr
    z = F.leaky_relu(sk + main, alpha)
    dZ = torch.randn(z.shape, generator=g)
    z.backward(dZ)

    rawg, xg = cl(raw).to(DEV), cl(xs).to(DEV)
    rows = Nn * T * H * W
    flat = rawg.view(rows, -1)
    part = torch.stack([flat.sum(0), (flat ** 2).sum(0)]).unsqueeze(0).contiguous()
    st = ops.bn_finalize(part, C, rows, gamma.to(DEV), beta.to(DEV))
    mv = ops.view(rawg, st[2], st[3], slope)
    gsg, bsg = gs.to(DEV), bs.to(DEV)
    sv = ops.view(xg, gsg, bsg, slope) if skip_is_view else ops.view(xg)
    zg = ops.residual_fwd(sv, mv, alpha, rawg, C)
    torch.cuda.synchronize()
    assert relerr(uncl(zg.cpu(), C), z.detach()) < 2e-5
    d_raw, dS, dgam, dbet = ops.bn_backward(cl(dZ).to(DEV), mv, st, C, skip=sv, alpha=alpha)
    torch.cuda.synchronize()
    assert relerr(uncl(d_raw.cpu(), C), rawr.grad) < 5e-5
    # dS is the gradient w.r.t. the (activated) skip operand
    if skip_is_view:
        pre = xs * gs.view(1, -1, 1, 1, 1) + bs.view(1, -1, 1, 1, 1)
        ref_dS = xr.grad / (gs.view(1, -1, 1, 1, 1) * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, slope)))
    else:
        ref_dS = xr.grad
    assert relerr(uncl(dS.cpu(), C), ref_dS) < 5e-5


def test_layout_and_pool_roundtrip():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 7, 4, 5, 6, generator=g)
    xg = ops.to_channels_last(x.to(DEV))
    assert torch.equal(xg.cpu(), cl(x))                       # bit exact
    assert torch.equal(ops.from_channels_last(xg, 7).cpu(), x)
    feat = ops.avgpool_fwd(xg, 7)
    assert relerr(feat.cpu(), x.mean(dim=(2, 3, 4))) < 1e-6
    df = torch.randn(3, 7, generator=g)
    dx = ops.avgpool_bwd(df.to(DEV), xg.shape)
    ref = (df / (4 * 5 * 6)).view(3, 1, 1, 1, 7).expand(3, 4, 5, 6, 7)
    assert relerr(dx[..., :7].cpu(), ref) < 1e-6
    assert float(dx[..., 7:].abs().max().cpu()) == 0.0


def test_losses_against_reference_fixture(golden_dir):
    from oracle import losses as ol
    gd = np.load(os.path.join(golden_dir, "losses.npz"))
    w = torch.from_numpy(gd["w"]).to(DEV)
    m = ol.ldam_margins([100, 2000], 0.5).to(DEV)
    cases = {"focal_g2": ("focal", w, None, 2.0), "focal_g0p5": ("focal", w, None, 0.5),
             "ldam_s30": ("ldam", w, m, 30.0), "ldam_s1_now": ("ldam", None, m, 1.0), "ce": ("ce", w, None, 0.0)}
    for B in (1, 8, 33):
        x = torch.from_numpy(gd[f"x{B}"]).to(DEV); y = torch.from_numpy(gd[f"y{B}"]).to(DEV)
        for name, (kind, cw, mm, gs) in cases.items():
            L, dl, pred = ops.softmax_loss(kind, x, y, cw, mm, gs)
            torch.cuda.synchronize()
            assert abs(float(L.cpu()) - float(gd[f"{name}/L{B}"])) <= 1e-5 * max(1.0, abs(float(gd[f"{name}/L{B}"]))), (name, B)
            assert relerr(dl.cpu(), torch.from_numpy(gd[f"{name}/g{B}"])) < 1e-4, (name, B)
            # bookkeeping: bit-exact argmax of softmax (src/train.py:70)
            assert torch.equal(pred.cpu(), torch.softmax(x.cpu(), 1).max(1)[1]), (name, B)

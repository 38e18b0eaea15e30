"""GPU: the persistent convolution kernels (csrc/conv_pers.hip: weights resident in LDS, one workgroup walking many
boxes, BatchNorm partial sums per workgroup) and the BatchNorm-backward reduction fused into the data gradient's
epilogue (md_conv_dgrad_bnred + md_bn_bwd_apply_g).

By default those kernels only take layers with at least one box per CU, which the small parity shapes never reach, so
the tests force a grid of a few workgroups with md_set_pers_grid (every workgroup then walks several boxes, the tensor
edge cuts boxes, the last round is ragged) and check against F.conv3d / autograd on the CPU.  Reference semantics:
Conv3dBlock forward/backward, src/models/R2Plus1D.py:44-58.  Tolerances as in test_conv_random_gpu.py: 3e-5 forward,
5e-5 gradients (relative to the tensor's max magnitude)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src import _native as N
    from src import ops

from tests.test_ops_gpu import cl, uncl, relerr

DEV = "cuda:0"

# name, Cin, Cout, kernel, stride, pad, (N,T,H,W)
CASES = [
    ("sp3x3", 32, 72, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 20, 18)),          # 5 column tiles: split 3 | 2 over the wave halves
    ("tmp3", 72, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 9, 10, 9)),            # 2 column tiles, 9 chunks per pixel
    ("stem", 3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3), (2, 3, 36, 40)),            # pixel-pair form
    ("tmp45", 45, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 21, 9, 8)),
    ("skip1x1", 32, 21, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 5, 18, 17)),
    ("tmp3s2", 115, 64, (3, 1, 1), (2, 1, 1), (1, 0, 0), (2, 9, 9, 9)),
    ("odd", 21, 64, (1, 1, 1), (2, 1, 1), (0, 0, 0), (2, 7, 11, 13)),
    ("k5", 24, 40, (1, 5, 3), (1, 1, 1), (0, 2, 1), (1, 2, 23, 19)),
    ("cls_s122", 20, 24, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 2, 131, 128)),     # residue-class data gradient (4 launches)
    ("cls_s211", 24, 16, (3, 1, 1), (2, 1, 1), (1, 0, 0), (2, 9, 64, 64)),
]


@pytest.fixture
def pers_grid():
    lib = N.lib()
    prev = lib.md_set_pers_grid(3)
    yield 3
    lib.md_set_pers_grid(prev)


def _bn_stats(y):
    """Train-mode BatchNorm statistics of a CPU (B,C,T,H,W) tensor -> mean, invstd (biased variance, eps 1e-5)."""
    m = y.double().mean(dim=(0, 2, 3, 4))
    v = y.double().var(dim=(0, 2, 3, 4), unbiased=False)
    return m.float(), (1.0 / torch.sqrt(v + 1e-5)).float()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_persistent_forward_and_data_gradient(case, pers_grid):
    name, Cin, Cout, k, s, p, (Nn, T, H, W) = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    x = torch.randn(Nn, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / np.sqrt(Cin * k[0] * k[1] * k[2])
    sc = torch.rand(Cin, generator=g) + 0.5
    sh = torch.randn(Cin, generator=g) * 0.3
    slope = 0.1
    d = ops.make_desc(Nn, T, H, W, Cin, Cout, k, s, p)
    used = 0
    for prologue in (False, True):
        xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
        a = F.leaky_relu(xr * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1), slope) if prologue else xr
        a.retain_grad()
        y = F.conv3d(a, wr, None, s, p)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        xg, dyg, wg = cl(x).to(DEV), cl(dy).to(DEV), w.to(DEV)
        Cp = xg.shape[-1]
        scp = torch.zeros(Cp); scp[:Cin] = sc; shp = torch.zeros(Cp); shp[:Cin] = sh
        scg, shg = scp.to(DEV), shp.to(DEV)                   # (the view holds raw pointers: keep the tensors alive)
        v = ops.view(xg, scg, shg, slope) if prologue else ops.view(xg)
        wf, wd = ops.pack_weights(d, wg)
        yg, part = ops.conv_fwd(d, v, wf, DEV, want_stats=True)
        used += int(part.shape[0] == pers_grid)
        dx = ops.conv_dgrad(d, dyg, wd)
        base = torch.randn(dx.shape, generator=g).to(DEV)
        base[..., Cin:] = 0
        dx2 = ops.conv_dgrad(d, dyg, wd, out=base.clone(), accumulate=True)
        torch.cuda.synchronize()
        tag = (name, prologue)
        assert relerr(uncl(yg.cpu(), Cout), y.detach()) < 3e-5, tag
        assert relerr(uncl(dx.cpu(), Cin), a.grad) < 5e-5, tag
        assert relerr(uncl((dx2 - base).cpu(), Cin), a.grad) < 5e-5, tag
        # BatchNorm partial sums: sum and sum of squares per output channel over the VALID pixels
        s1 = part[:, 0, :Cout].double().sum(0).cpu(); s2 = part[:, 1, :Cout].double().sum(0).cpu()
        r1 = y.detach().double().sum(dim=(0, 2, 3, 4)); r2 = (y.detach().double() ** 2).sum(dim=(0, 2, 3, 4))
        assert float((s1 - r1).abs().max()) < 1e-3 * max(1.0, float(r1.abs().max())), tag
        assert float((s2 - r2).abs().max()) < 1e-4 * max(1.0, float(r2.abs().max())), tag
        if yg.shape[-1] > Cout:
            assert float(yg[..., Cout:].abs().max()) == 0.0, tag
        if dx.shape[-1] > Cin:
            assert float(dx[..., Cin:].abs().max()) == 0.0, tag
    if name in ("sp3x3", "tmp3", "stem", "tmp45"):
        assert used == 2, f"{name}: the persistent forward kernel was expected to take this geometry"


FUSE_CASES = [c for c in CASES if c[0] in ("sp3x3", "tmp3", "tmp45", "cls_s122", "cls_s211", "k5")]


@pytest.mark.parametrize("case", FUSE_CASES, ids=[c[0] for c in FUSE_CASES])
@pytest.mark.parametrize("accumulate", [False, True])
def test_fused_batchnorm_backward_reduction(case, accumulate, pers_grid):
    """unit P (conv -> BN(train) -> LeakyReLU) feeds conv Q.  Backward of P's BatchNorm given dY of Q: the fused path
    (Q's data gradient writes g and the partial sums, then finalize + apply_g) against autograd on the CPU and against
    the unfused kernel sequence (data gradient, reduce, finalize, apply)."""
    name, Cin, Cout, k, s, p, (Nn, T, H, W) = case
    g = torch.Generator().manual_seed(7 + sum(map(ord, name)))
    yP = torch.randn(Nn, Cin, T, H, W, generator=g) * 1.5 + 0.2           # raw output of unit P (= BN input)
    gamma = torch.rand(Cin, generator=g) + 0.5
    beta = torch.randn(Cin, generator=g) * 0.3
    w = torch.randn(Cout, Cin, *k, generator=g) / np.sqrt(Cin * k[0] * k[1] * k[2])
    slope = 0.01
    yr = yP.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    a = F.leaky_relu(F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5), slope)
    out = F.conv3d(a, w, None, s, p)
    dy = torch.randn(out.shape, generator=g)
    extra = torch.randn(a.shape, generator=g) * 0.5 if accumulate else None   # gradient reaching `a` from a second consumer
    loss = (out * dy).sum() + ((a * extra).sum() if accumulate else 0.0)
    loss.backward()

    d = ops.make_desc(Nn, T, H, W, Cin, Cout, k, s, p)
    mean, invstd = _bn_stats(yP)
    Cp = ops.cpad(Cin)
    st = torch.zeros(4, Cp)
    st[0, :Cin] = mean; st[1, :Cin] = invstd; st[2, :Cin] = gamma * invstd; st[3, :Cin] = beta - mean * gamma * invstd
    st = st.to(DEV)
    yg, dyg = cl(yP).to(DEV), cl(dy).to(DEV)
    _, wd = ops.pack_weights(d, w.to(DEV))
    yv = ops.view(yg, st[2], st[3], slope)
    base = cl(extra).to(DEV) if accumulate else None

    fused = ops.conv_dgrad_bnred(d, dyg, wd, yv, st, out=None if base is None else base.clone(), accumulate=accumulate)
    if fused is None:
        assert Cin > 48, "a data gradient with <= 3 destination-channel tiles is expected to have the fused form"
        pytest.skip("no fused form instantiated for more than 3 destination-channel tiles (register budget)")
    gq, part = fused
    d_raw, dgamma, dbeta = ops.bn_backward_from_g(gq, part, yv, st, Cin)
    # unfused sequence on the same inputs
    dA = ops.conv_dgrad(d, dyg, wd, out=None if base is None else base.clone(), accumulate=accumulate)
    d_raw0, _, dgamma0, dbeta0 = ops.bn_backward(dA, yv, st, Cin)
    torch.cuda.synchronize()
    tag = (name, accumulate)
    assert relerr(uncl(d_raw.cpu(), Cin), yr.grad) < 5e-5, tag
    assert relerr(dgamma.cpu(), gr.grad) < 5e-5 and relerr(dbeta.cpu(), br.grad) < 5e-5, tag
    assert relerr(d_raw, d_raw0) < 1e-5 and relerr(dgamma, dgamma0) < 1e-5 and relerr(dbeta, dbeta0) < 1e-5, tag
    if d_raw.shape[-1] > Cin:
        assert float(d_raw[..., Cin:].abs().max()) == 0.0, tag


PATCH_FUSE_CASES = CASES + [
    ("wide72", 72, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 6, 12, 12)),          # 5 destination tiles (the 64x64 stage's 72-channel tensors)
    ("wide144", 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 5, 8, 8)),          # 9 tiles: column halves of the eight-wave form
    ("wide115s", 115, 64, (3, 1, 1), (2, 1, 1), (1, 0, 0), (2, 9, 9, 9)),         # strided (per-tap test form), 8 tiles, Cp % 8 == 4
    ("tiny288", 128, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 8, 8)),         # few boxes: the 64-pixel form
]


@pytest.mark.parametrize("case", [c for c in PATCH_FUSE_CASES if c[0] != "stem"], ids=[c[0] for c in PATCH_FUSE_CASES if c[0] != "stem"])
@pytest.mark.parametrize("accumulate", [False, True])
def test_fused_reduction_in_the_per_box_kernels(case, accumulate):
    """Round 3: the per-box data-gradient kernels (k_conv_patch: classic, eight-wave, 64-pixel, strided and residue-class forms)
    carry the fused BatchNorm-backward reduction too, for ANY number of destination tiles -- default grid (no persistent override),
    so geometries without a persistent form take them.  Same references as the persistent test: autograd on the CPU and the unfused
    kernel sequence."""
    name, Cin, Cout, k, s, p, (Nn, T, H, W) = case
    g = torch.Generator().manual_seed(17 + sum(map(ord, name)))
    yP = torch.randn(Nn, Cin, T, H, W, generator=g) * 1.5 + 0.2
    gamma = torch.rand(Cin, generator=g) + 0.5
    beta = torch.randn(Cin, generator=g) * 0.3
    w = torch.randn(Cout, Cin, *k, generator=g) / np.sqrt(Cin * k[0] * k[1] * k[2])
    slope = 0.01
    yr = yP.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    a = F.leaky_relu(F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5), slope)
    out = F.conv3d(a, w, None, s, p)
    dy = torch.randn(out.shape, generator=g)
    extra = torch.randn(a.shape, generator=g) * 0.5 if accumulate else None
    ((out * dy).sum() + ((a * extra).sum() if accumulate else 0.0)).backward()
    d = ops.make_desc(Nn, T, H, W, Cin, Cout, k, s, p)
    mean, invstd = _bn_stats(yP)
    Cp = ops.cpad(Cin)
    st = torch.zeros(4, Cp)
    st[0, :Cin] = mean; st[1, :Cin] = invstd; st[2, :Cin] = gamma * invstd; st[3, :Cin] = beta - mean * gamma * invstd
    st = st.to(DEV)
    yg, dyg = cl(yP).to(DEV), cl(dy).to(DEV)
    _, wd = ops.pack_weights(d, w.to(DEV))
    yv = ops.view(yg, st[2], st[3], slope)
    base = cl(extra).to(DEV) if accumulate else None
    fused = ops.conv_dgrad_bnred(d, dyg, wd, yv, st, out=None if base is None else base.clone(), accumulate=accumulate)
    assert fused is not None, name
    gq, part = fused
    d_raw, dgamma, dbeta = ops.bn_backward_from_g(gq, part, yv, st, Cin)
    dA = ops.conv_dgrad(d, dyg, wd, out=None if base is None else base.clone(), accumulate=accumulate)
    d_raw0, _, dgamma0, dbeta0 = ops.bn_backward(dA, yv, st, Cin)
    torch.cuda.synchronize()
    tag = (name, accumulate)
    assert relerr(uncl(d_raw.cpu(), Cin), yr.grad) < 5e-5, tag
    assert relerr(dgamma.cpu(), gr.grad) < 5e-5 and relerr(dbeta.cpu(), br.grad) < 5e-5, tag
    assert relerr(d_raw, d_raw0) < 1e-5 and relerr(dgamma, dgamma0) < 1e-5 and relerr(dbeta, dbeta0) < 1e-5, tag
    if d_raw.shape[-1] > Cin:
        assert float(d_raw[..., Cin:].abs().max()) == 0.0, tag


def test_trunk_with_persistent_kernels_against_the_oracle():
    """Whole R(2+1)D classifier, forward + loss + backward, on a clip large enough that the default plan picks the
    persistent kernels (and the fused reduction where it exists) for the 64x64-resolution units (448 boxes >= 2 x 128):
    logits and loss within 1e-4 of the fp64 oracle; every parameter gradient within 1e-3 (relative L2) of the fp64 oracle
    evaluated on the activation pattern the HIP forward took (tests/kink_util.py: the sign flips against the oracle's own
    pattern are listed and must be few and within rounding error of zero)."""
    from oracle import losses as ol, r2plus1d as orc, step as ostep
    from src.loss import FocalLoss
    from src.models.R2Plus1D import R2Plus1DClassifier
    from tests import kink_util as ku
    layers, alpha = [1, 1, 1, 1], 0.01
    B, T, S, seed = 2, 7, 128, 5
    params, bufs = orc.synth_state(layers, seed, alpha)
    x = orc.synth_clip(B, T, S, seed); y = orc.synth_labels(B, seed)
    p64 = {k: v.double() for k, v in params.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in bufs.items()}
    w = torch.ones(2, dtype=torch.float64)
    logits, loss, _ = ostep.r2plus1d_loss_and_grads(x.double(), y, p64, b64, layers, alpha, lambda o, t: ol.focal_loss(o, t, w, 2.0))
    m = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=layers, alpha=alpha)
    params, bufs = orc.synth_state(layers, seed, alpha)
    sd = dict(params); sd.update(bufs)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    lg = m(x.to(DEV))
    lossg = FocalLoss(weight=torch.ones(2), gamma=2.0)(lg, y.to(DEV))
    lossg.backward()
    torch.cuda.synchronize()
    assert relerr(lg.detach().cpu(), logits.detach()) < 1e-4
    assert abs(lossg.item() - float(loss)) < 1e-4 * max(1.0, abs(float(loss)))
    rep = ku.gradient_report(m, x, y, layers, alpha, seed, torch.ones(2), 2.0, DEV)
    for name, idx, vh, vo, rms in rep["flips"][:12]:
        print(f"  sign flip: {name}[{idx}]  hip {vh:+.3e}  fp64 oracle {vo:+.3e}  (tensor rms {rms:.3e})")
    print("gradients vs fp64 oracle: own pattern %.2e, HIP pattern worst %.2e (%s) median %.2e, flips %d" % (
        rep["worst_own"], rep["worst_pattern"], rep["worst_name"], rep["median_pattern"], len(rep["flips"])))
    assert len(rep["flips"]) <= rep["max_flips"], (len(rep["flips"]), rep["elements"])
    for name, idx, vh, vo, rms in rep["flips"]:
        assert abs(vh) <= 1e-4 * rms and abs(vo) <= 1e-4 * rms, (name, idx, vh, vo, rms)
    assert rep["worst_pattern"] < 1e-3, rep


SPLIT_CASES = [c for c in CASES if c[0] != "stem"] + [
    ("wide144", 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 3, 12, 12)),      # weight gradient split over several column groups
    ("t144", 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 5, 8, 8)),
    ("big", 32, 72, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 6, 40, 36)),            # enough boxes for the prefetching weight-gradient kernels
]


@pytest.mark.parametrize("case", SPLIT_CASES, ids=[c[0] for c in SPLIT_CASES])
@pytest.mark.parametrize("persistent", [False, True], ids=["per_box", "persistent"])
def test_presplit_gradient_format_is_bit_identical(case, persistent):
    """d_raw written by the apply pass in the pre-split bf16 format (md_bn_bwd_apply_fmt, split_out) and consumed by the data
    gradient and the weight gradient with plain copies must give the SAME BITS as the fp32 tensor split while it is staged:
    the halves come from the same split8() arithmetic."""
    name, Cin, Cout, k, s, p, (Nn, T, H, W) = case
    lib = N.lib()
    prev = lib.md_set_pers_grid(3 if persistent else 0)
    try:
        g = torch.Generator().manual_seed(11 + sum(map(ord, name)))
        d = ops.make_desc(Nn, T, H, W, Cin, Cout, k, s, p)
        if ops.cpad(Cout) % 8:
            pytest.skip("the pre-split format needs a channel pitch that is a multiple of 8")
        x = torch.randn(Nn, Cin, T, H, W, generator=g)
        w = torch.randn(Cout, Cin, *k, generator=g) / np.sqrt(Cin * k[0] * k[1] * k[2])
        y = F.conv3d(x, w, None, s, p)                                   # raw output of the unit: BatchNorm input
        dA = torch.randn(y.shape, generator=g) * torch.logspace(-6, 2, y.shape[1]).view(1, -1, 1, 1, 1)    # wide dynamic range
        gamma = torch.rand(Cout, generator=g) + 0.5; beta = torch.randn(Cout, generator=g) * 0.3
        mean, invstd = _bn_stats(y)
        Cp = ops.cpad(Cout)
        st = torch.zeros(4, Cp); st[0, :Cout] = mean; st[1, :Cout] = invstd
        st[2, :Cout] = gamma * invstd; st[3, :Cout] = beta - mean * gamma * invstd
        st = st.to(DEV)
        coef = (torch.randn(2, Cp, generator=g) * 0.01).to(DEV); coef[:, Cout:] = 0
        yg, dAg, xg = cl(y).to(DEV), cl(dA).to(DEV), cl(x).to(DEV)
        yv = ops.view(yg, st[2], st[3], 0.01)
        d32 = ops.bn_apply_fmt(dAg, yv, st, coef, Cout, split_out=False)
        dsp = ops.bn_apply_fmt(dAg, yv, st, coef, Cout, split_out=True)
        # the split tensor holds hi = bf16(x) and lo ~ bf16(x - hi) of the fp32 one, chunk by chunk: 16 significant bits
        hi_lo = dsp.view(torch.bfloat16).view(-1, 2, 8).float()
        ref = d32.view(-1, 8)
        assert torch.equal(hi_lo[:, 0], ref.bfloat16().float())
        assert float(((hi_lo[:, 0] + hi_lo[:, 1] - ref).abs() / ref.abs().clamp_min(1e-30)).max()) < 2.0 ** -15
        _, wd = ops.pack_weights(d, w.to(DEV))
        dx32 = ops.conv_dgrad_fmt(d, d32, False, wd)
        dxsp = ops.conv_dgrad_fmt(d, dsp, True, wd)
        assert torch.equal(dx32, dxsp), name
        form = lib.md_set_wgrad_form(1)        # both launches on the kernel form that reads either format
        try:
            dw32 = ops.conv_wgrad_fmt(d, ops.view(xg), d32, False)
            dwsp = ops.conv_wgrad_fmt(d, ops.view(xg), dsp, True)
            torch.cuda.synchronize()
        finally:
            lib.md_set_wgrad_form(form)
        assert torch.equal(dw32, dwsp), name
    finally:
        lib.md_set_pers_grid(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,k,pad,shape", [(32, 72, (1, 3, 3), (0, 1, 1), (2, 5, 32, 32)), (72, 32, (3, 1, 1), (1, 0, 0), (2, 6, 24, 24)),
                                                  (115, 64, (3, 1, 1), (1, 0, 0), (1, 5, 16, 16)), (64, 144, (1, 3, 3), (0, 1, 1), (1, 3, 20, 20)),
                                                  (21, 64, (1, 1, 1), (0, 0, 0), (2, 4, 16, 16))])
def test_weight_gradient_from_the_presplit_activation_copy_is_bit_identical(cin, cout, k, pad, shape):
    """md_bn_act_split + md_conv_wgrad_fmt2(x_split): the weight gradient that stages a pre-activated, pre-split bf16 copy of its
    input by plain copy against the one that applies BatchNorm-on-read and splits inside the kernel -- same bits (channel counts
    with Cp % 8 == 4 and Cp < 16-channel k tiles included), also for a materialised input without BatchNorm view."""
    from src import ops
    N_, T, H, W = shape
    g = torch.Generator().manual_seed(cin * 7 + cout)
    d = ops.make_desc(N_, T, H, W, cin, cout, k, (1, 1, 1), pad)
    Cp = ops.cpad(cin)
    x = torch.zeros(N_, T, H, W, Cp); x[..., :cin] = torch.randn(N_, T, H, W, cin, generator=g)
    sc = torch.zeros(Cp); sh = torch.zeros(Cp); sc[:cin] = torch.rand(cin, generator=g) + 0.5; sh[:cin] = torch.randn(cin, generator=g) * 0.3
    dy = torch.zeros(N_, d.To, d.Ho, d.Wo, ops.cpad(cout)); dy[..., :cout] = torch.randn(N_, d.To, d.Ho, d.Wo, cout, generator=g) * 1e-3
    x, sc, sh, dy = x.cuda(), sc.cuda(), sh.cuda(), dy.cuda()
    rows = N_ * T * H * W
    form = N.lib().md_set_wgrad_form(1)            # the pre-split formats are read by the first kernel form only
    try:
        for with_bn in (True, False):
            v = ops.view(x, sc, sh, 0.01) if with_bn else ops.view(x)
            ref = ops.conv_wgrad(d, v, dy)
            xs = ops.bn_act_split(v, rows, cin, x.device)
            got = ops.conv_wgrad_xsplit(d, xs, dy)
            torch.cuda.synchronize()
            assert torch.equal(ref, got), (with_bn, float((ref - got).abs().max()))
    finally:
        N.lib().md_set_wgrad_form(form)

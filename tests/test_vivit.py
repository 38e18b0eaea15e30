"""Scope rows a9-a11: ViViT / ViViTEncoder (src/models/ViViT.py:12-299).
CPU: the oracle restatement (einops patterns written out) against the fixture recorded from the reference (2e-5).
GPU: the native modules against the same fixture: outputs within 1e-3 of their scale, input and parameter gradients within 3e-3
relative L2 (both arithmetic modes); the BASELINE cfg3 shape (4,3,21,224,224) runs forward + backward with finite results and
the documented shapes; ELU and the residual-stream LayerNorm against PyTorch on the CPU."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vivit as ov

BASE = dict(image_size=32, patch_size=8, n_frames=5, dim=32, depth=2, n_heads=2, in_channels=3, d_head=16, dropout=0.0,
            embedd_dropout=0.0, scale_dim=2)
CASES = {"cls": ("ViViT", dict(n_classes=2, pool="cls", alpha=0.7)), "enc": ("ViViTEncoder", dict(pool="mean"))}


def _load(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "vivit.npz"))
    pre = tag + "/sd/"
    return g, {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}


@pytest.mark.parametrize("tag", ["cls", "enc"])
def test_oracle_matches_reference_fixture(golden_dir, tag):
    g, sd = _load(golden_dir, tag)
    kw = CASES[tag][1]
    out = ov.vivit_forward(torch.from_numpy(g[tag + "/x"]), sd, 8, 2, 2, kw["pool"], 3, kw.get("alpha", 1.0), with_mlp=(tag == "cls"))
    ref = torch.from_numpy(g[tag + "/out"])
    assert float((out - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def _relerr(a, b):
    return float((a.double() - b.double()).norm() / max(1e-12, float(b.double().norm())))


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False], ids=["exact_fp32", "split"])
@pytest.mark.parametrize("tag", ["cls", "enc"])
def test_native_module_matches_reference_fixture(golden_dir, tag, exact):
    from src import ops
    from src.models import ViViT as V
    g, sd = _load(golden_dir, tag)
    name, kw = CASES[tag]
    m = getattr(V, name)(**BASE, **kw)
    m.load_state_dict(sd, strict=True)
    m.cuda().train()
    x = torch.from_numpy(g[tag + "/x"]).cuda().requires_grad_(True)
    ops.set_exact_fp32(exact)
    try:
        out = m(x)
        out.backward(torch.from_numpy(g[tag + "/dout"]).cuda())
        torch.cuda.synchronize()
    finally:
        ops.set_exact_fp32(False)
    ref = torch.from_numpy(g[tag + "/out"])
    assert float((out.detach().cpu() - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()))
    assert _relerr(x.grad.cpu(), torch.from_numpy(g[tag + "/dx"])) < 3e-3
    for k, p in m.named_parameters():
        r = torch.from_numpy(g[tag + "/grad/" + k])
        assert p.grad is not None, k
        assert _relerr(p.grad.cpu(), r) < 3e-3, (k, _relerr(p.grad.cpu(), r))
    # the channels-first clip layout is accepted as well (ViViT.py:174-175) and encode() is the forward without the mlp
    m.eval()
    with torch.no_grad():
        x5 = x.detach()
        a = m(x5); b = m(x5.permute(0, 2, 1, 3, 4).contiguous())
        assert torch.equal(a, b)
        if tag == "cls":
            assert tuple(m.encode(x5).shape) == (3, 32)


@pytest.mark.gpu
def test_baseline_cfg3_shape_runs():
    """BASELINE.json configs[2]: (B=4, T=21, 3, 224, 224), patch 16, dim 128, depth 2, heads 4, d_head 64, scale_dim 8, pool mean."""
    from src.models.ViViT import ViViT
    torch.manual_seed(0)
    m = ViViT(image_size=224, patch_size=16, n_frames=21, n_classes=2, dim=128, depth=2, n_heads=4, pool="mean", in_channels=3,
              d_head=64, dropout=0.1, embedd_dropout=0.1, scale_dim=8).cuda().train()
    x = torch.randn(4, 3, 21, 224, 224, device="cuda")
    out = m(x)
    assert tuple(out.shape) == (4, 2)
    out.sum().backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    for k, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    assert float(m.space_transformer.layers[0][0].fn.to_qkv.weight.grad.abs().max()) > 0


@pytest.mark.gpu
def test_baseline_cfg3_against_the_oracle():
    """BASELINE.json configs[2] at its real size (B=4, T=21, 3, 224, 224; patch 16, dim 128, depth 2, heads 4, d_head 64,
    scale_dim 8, pool mean), dropout 0: logits within 1e-3 and EVERY parameter gradient within 3e-3 relative L2 of
    oracle/vivit.py on the CPU (fp32) from the same weights -- the gather-GEMM patch embedding, the log-sum-exp attention at
    197 tokens and the eight-wave Linear kernels end to end (reference src/models/ViViT.py:141-223)."""
    from src.models.ViViT import ViViT
    torch.set_num_threads(16)
    torch.manual_seed(3)
    m = ViViT(image_size=224, patch_size=16, n_frames=21, n_classes=2, dim=128, depth=2, n_heads=4, pool="mean", in_channels=3,
              d_head=64, dropout=0.0, embedd_dropout=0.0, scale_dim=8)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 3, 21, 224, 224, generator=g)
    dl = torch.tensor([[0.4, -0.6], [-0.3, 0.2], [0.5, 0.1], [-0.2, -0.4]])
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = ov.vivit_forward(x, leaves, 16, 2, 4, "mean", 3, 1.0, with_mlp=True)
    (ref * dl).sum().backward()
    m = m.cuda().train()
    out = m(x.cuda())
    out.backward(dl.cuda())
    torch.cuda.synchronize()
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 1e-3 * max(1.0, float(ref.detach().abs().max()))
    worst, checked = (0.0, ""), 0
    for k, p in m.named_parameters():
        r = leaves[k].grad
        assert p.grad is not None and r is not None, k
        e = _relerr(p.grad.cpu(), r)
        worst = max(worst, (e, k))
        assert e < 3e-3, (k, e)
        checked += 1
    assert checked >= 10
    for must in ("to_patch_embedding.1.weight", "pos_embedding", "space_transformer.layers.0.0.fn.to_qkv.weight",
                 "space_transformer.layers.0.1.fn.net.0.weight", "space_transformer.layers.0.1.fn.net.3.weight"):
        assert must in leaves, must
    print("cfg3 full size: worst parameter-gradient relative L2 %.2e (%s) over %d tensors" % (worst[0], worst[1], checked))


@pytest.mark.gpu
def test_elu_and_residual_layernorm_match_torch():
    from src.models._unit import EluFunction, ResidualLayerNormFunction
    torch.manual_seed(9)
    x = torch.linspace(-8, 8, 2001)
    for alpha in (1.0, 0.7):
        xr = x.clone().requires_grad_(True); yr = F.elu(xr, alpha); yr.backward(torch.ones_like(x))
        xg = x.cuda().requires_grad_(True); yg = EluFunction.apply(xg, alpha); yg.backward(torch.ones_like(xg))
        assert float((yg.detach().cpu() - yr.detach()).abs().max()) < 2e-6 and float((xg.grad.cpu() - xr.grad).abs().max()) < 2e-6
    for rows in (40, 3000):                                   # single-pass and chunked parameter-gradient reductions
        a = torch.randn(rows, 96); b = torch.randn(rows, 96); ga = torch.rand(96) + 0.5; be = torch.randn(96)
        ds = torch.randn(rows, 96); dh = torch.randn(rows, 96)
        ar, br, gr, ber = (t.clone().requires_grad_(True) for t in (a, b, ga, be))
        s = ar + br; h = F.layer_norm(s, (96,), gr, ber, 1e-5)
        torch.autograd.backward((s, h), (ds, dh))
        ag, bg, gg, beg = (t.cuda().requires_grad_(True) for t in (a, b, ga, be))
        s2, h2 = ResidualLayerNormFunction.apply(ag, bg, gg, beg, 1e-5)
        torch.autograd.backward((s2, h2), (ds.cuda(), dh.cuda()))
        for x_, y_ in ((s2.detach(), s.detach()), (h2.detach(), h.detach()), (ag.grad, ar.grad), (bg.grad, br.grad), (gg.grad, gr.grad),
                       (beg.grad, ber.grad)):
            assert float((x_.cpu() - y_).abs().max()) < 3e-5 * max(1.0, float(y_.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,din,dout", [(1000, 768, 128), (16548, 1024, 128), (777, 128, 768), (300, 1024, 130), (168, 1000, 36)])
def test_wide_linear_matches_fp64(rows, din, dout):
    """Linears with more than 320 source channels in the forward (or in the data gradient) take the K-streaming split-precision
    GEMM (k_linear_split); checked against an fp64 matmul: forward and both gradients within 2e-5 relative L2 (fp32-level)."""
    from src.models._unit import linear_wb
    torch.manual_seed(rows + din)
    x = torch.randn(rows, din); w = torch.randn(dout, din) / din ** 0.5; b = torch.randn(dout); dy = torch.randn(rows, dout)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = xr @ wr.t() + br
    ref.backward(dy.double())
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    out = linear_wb(xg, wg, bg)
    out.backward(dy.cuda())
    torch.cuda.synchronize()
    rel = lambda a, r: float((a.double().cpu() - r).norm() / r.norm())
    assert rel(out.detach(), ref.detach()) < 2e-5
    assert rel(xg.grad, xr.grad) < 2e-5
    assert rel(wg.grad, wr.grad) < 2e-5
    assert rel(bg.grad, br.grad) < 2e-5


@pytest.mark.gpu
def test_single_head_identity_projection_and_long_sequences():
    """Attention with n_heads == 1 and d_head == dim has no output projection (ViViT.py:60,66-69: nn.Identity); checked against
    the oracle restatement on the module's own random state.  Sequences beyond the matrix-core kernels' 256 tokens fall back to the
    row-blocked kernels (checked against PyTorch), beyond their LDS budget the call fails loudly."""
    from src.models.ViViT import ViViT
    from src.models._unit import AttentionFunction
    torch.manual_seed(12)
    m = ViViT(image_size=16, patch_size=8, n_frames=3, n_classes=2, dim=16, depth=1, n_heads=1, pool="cls", in_channels=3, d_head=16,
              dropout=0.0, embedd_dropout=0.0, scale_dim=2).cuda().train()
    assert isinstance(m.space_transformer.layers[0][0].fn.to_out, torch.nn.Identity)
    x = torch.randn(2, 3, 3, 16, 16)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = ov.vivit_forward(x, sd, 8, 1, 1, "cls", 3, 1.0)
    out = m(x.cuda())
    assert float((out.detach().cpu() - ref).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()))
    out.sum().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    S, B, D, H = 300, 1, 32, 2
    qkv = torch.randn(B, S, 3 * D)
    q, k, v = (t.reshape(B, S, H, D // H).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    ref = (torch.softmax(q @ k.transpose(2, 3) / (D // H) ** 0.5, dim=-1) @ v).transpose(1, 2).reshape(B, S, D)
    got = AttentionFunction.apply(qkv.cuda(), None, H, None, True)
    assert float((got.cpu() - ref).abs().max()) < 3e-6
    with pytest.raises(RuntimeError):
        AttentionFunction.apply(torch.randn(1, 1200, 3 * D).cuda(), None, H, None, True)


@pytest.mark.gpu
@pytest.mark.parametrize("N,C", [(4, 8192), (3, 5000), (2100, 96), (4, 100)])
def test_bias_vector_gradient_paths(N, C):
    """md_channel_bias_* with L = 1 in its three reduction forms: one thread per entry (a table broadcast over few rows: ViViT's
    positional embedding), two-level column sums (many rows: Linear biases over all tokens), one workgroup per channel (small)."""
    from src.models._unit import _ChannelBias
    torch.manual_seed(N + C)
    x = torch.randn(N, C, 1); b = torch.randn(C); d = torch.randn(N, C, 1)
    xg, bg = x.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    out = _ChannelBias.apply(xg, bg)
    out.backward(d.cuda())
    assert torch.equal(out.detach().cpu(), x + b[None, :, None])
    ref = d.double().sum(dim=(0, 2))
    assert float((bg.grad.cpu().double() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max())) * max(1.0, N ** 0.5)
    assert torch.equal(xg.grad.cpu(), d)


@pytest.mark.gpu
def test_residule_prenorm_feedforward_modules_standalone():
    """The small wrapper classes of ViViT.py (:12-47) used on their own, as a caller composing a custom Transformer would:
    Residule(PreNorm(dim, FeedForward)) == ff(norm(x)) + x, forward and backward, against PyTorch on the CPU."""
    from src.models.ViViT import FeedForward, PreNorm, Residule
    torch.manual_seed(3)
    blk = Residule(PreNorm(24, FeedForward(24, 40, dropout=0.0)))
    with torch.no_grad():
        blk.fn.norm.weight.uniform_(0.5, 1.5); blk.fn.norm.bias.normal_(0, 0.3)
    x = torch.randn(5, 7, 24); d = torch.randn(5, 7, 24)
    xr = x.clone().requires_grad_(True)
    ff = blk.fn.fn.net
    h = F.layer_norm(xr, (24,), blk.fn.norm.weight, blk.fn.norm.bias, 1e-5)
    ref = F.linear(F.gelu(F.linear(h, ff[0].weight, ff[0].bias)), ff[3].weight, ff[3].bias) + xr
    ref.backward(d)
    ref_grads = {k: p.grad.clone() for k, p in blk.named_parameters()}
    blk.zero_grad()
    blk.cuda().train()
    xg = x.cuda().requires_grad_(True)
    out = blk(xg)
    out.backward(d.cuda())
    assert float((out.detach().cpu() - ref.detach()).abs().max()) < 2e-5
    assert float((xg.grad.cpu() - xr.grad).abs().max()) < 5e-5
    for k, p in blk.named_parameters():
        assert float((p.grad.cpu() - ref_grads[k]).abs().max()) <= 1e-4 * max(1.0, float(ref_grads[k].abs().max())), k


@pytest.mark.gpu
@pytest.mark.parametrize("b,t,c,S,p,dim,perm", [(2, 3, 3, 224, 16, 128, True), (1, 2, 3, 64, 8, 48, False), (2, 2, 1, 32, 32, 20, True)])
def test_fused_patch_embedding_matches_fp64(b, t, c, S, p, dim, perm):
    """md_patch_embed_* (gather-GEMM with bias, space token and positional table in the epilogue; weight gradient that gathers the
    patches again) against the reference's formulation in fp64 -- 'b t c (h p1) (w p2) -> b t (h w) (p1 p2 c)', Linear, token
    concatenation, positional add (ViViT.py:141-148, 175-184) -- at the cfg3 geometry and two others, on a (b,t,c,H,W) clip and
    on the permuted view of a (b,c,t,H,W) clip (read in place): output 2e-5, every gradient 3e-5 relative L2."""
    from src.models._unit import PatchEmbedFunction
    torch.manual_seed(b * 100 + p)
    n = (S // p) ** 2
    clip = torch.randn(b, c, t, S, S) * 40 if perm else torch.randn(b, t, c, S, S) * 40
    w = torch.randn(dim, p * p * c) / (p * p * c) ** 0.5 / 40; bias = torch.randn(dim); pos = torch.randn(t, n + 1, dim); tok = torch.randn(dim)
    gout = torch.randn(b * t, n + 1, dim)
    # fp64 reference, the reference's own order of operations
    x64 = (clip.permute(0, 2, 1, 3, 4) if perm else clip).double().requires_grad_(True)
    w64, b64, p64, t64 = (v.double().requires_grad_(True) for v in (w, bias, pos, tok))
    patches = x64.reshape(b, t, c, S // p, p, S // p, p).permute(0, 1, 3, 5, 4, 6, 2).reshape(b, t, n, p * p * c)
    e = patches @ w64.t() + b64
    e = torch.cat((t64.view(1, 1, 1, dim).expand(b, t, 1, dim), e), dim=2) + p64.unsqueeze(0)
    ref = e.reshape(b * t, n + 1, dim)
    ref.backward(gout.double())
    xg = clip.cuda().requires_grad_(True)
    wg, bg, pg, tg = (v.cuda().requires_grad_(True) for v in (w, bias, pos, tok))
    xin = xg.permute(0, 2, 1, 3, 4) if perm else xg
    w_perm = wg.view(dim, p, p, c).permute(0, 3, 1, 2).reshape(dim, c * p * p)
    out = PatchEmbedFunction.apply(xin, w_perm, bg, pg, tg, p)
    out.backward(gout.cuda())
    torch.cuda.synchronize()
    rel = lambda a, r: float((a.double().cpu() - r).norm() / r.norm())
    assert rel(out.detach(), ref.detach()) < 2e-5
    gx64 = x64.grad.permute(0, 2, 1, 3, 4) if perm else x64.grad
    for name, a, r in (("w", wg.grad, w64.grad), ("bias", bg.grad, b64.grad), ("pos", pg.grad, p64.grad), ("token", tg.grad, t64.grad),
                       ("x", xg.grad, gx64)):
        assert rel(a, r) < 3e-5, (name, rel(a, r))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,use_mask", [(0, True), (0, False), (1, True)])
def test_fused_bias_gelu_dropout_is_bit_identical_to_the_three_passes(kind, use_mask):
    """md_bias_gelu_drop (the FeedForward's hidden activation in one pass each way) against md_channel_bias_fwd -> md_gelu ->
    md_mask_scale and their backward passes: same arithmetic in the same order, so outputs, input gradient and bias gradient agree
    bit for bit."""
    from src.models._unit import BiasGeluDropFunction, GeluFunction, _ChannelBias, _MaskScale
    g = torch.Generator().manual_seed(31 + kind)
    x = torch.randn(777, 1024, generator=g).cuda(); b = torch.randn(1024, generator=g).cuda()
    mask = (torch.rand(777, 1024, generator=g) > 0.1).float().cuda() if use_mask else None
    dout = torch.randn(777, 1024, generator=g).cuda()
    x1, b1 = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y1 = BiasGeluDropFunction.apply(x1, b1, mask, 1.0 / 0.9, kind)
    y1.backward(dout)
    x2, b2 = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    h = GeluFunction.apply(_ChannelBias.apply(x2[:, :, None], b2)[:, :, 0], kind)
    y2 = _MaskScale.apply(h, mask, 1.0 / 0.9) if use_mask else h
    y2.backward(dout)
    assert torch.equal(y1, y2) and torch.equal(x1.grad, x2.grad) and torch.equal(b1.grad, b2.grad)

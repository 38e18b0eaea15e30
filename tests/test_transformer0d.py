"""Scope row a12: Transformer / TransformerEncoder 0D encoder (src/models/transformer.py:10-154).
CPU: the oracle restatement (nn.TransformerEncoderLayer written out) against the fixture recorded from the reference (logits
2e-5, running statistics 1e-6).  GPU: the native module against the same fixture: logits within 1e-3 of their scale, input and
parameter gradients within 3e-3 relative L2 (both arithmetic modes), running statistics 1e-4; biases in front of the
BatchNorm have analytically zero gradients and are bounded on both sides instead.  Op-level: attention / LayerNorm / GELU against
PyTorch on the CPU, incl. the causal -inf mask and the attention-dropout factors."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import transformer0d as ot

CFG = dict(n_features=18, kernel_size=5, feature_dims=64, max_len=21, n_layers=2, n_heads=4, dim_feedforward=96, dropout=0.0,
           cls_dims=32, n_classes=2)
ZERO = ("encoder.filter.1.bias",)


def _load(golden_dir):
    g = np.load(os.path.join(golden_dir, "transformer0d.npz"))
    return g, {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}


def test_oracle_matches_reference_fixture(golden_dir):
    g, sd = _load(golden_dir)
    sd = {k: v.clone() for k, v in sd.items()}
    out = ot.transformer0d_forward(torch.from_numpy(g["x"]), sd, 2, 4, 5, True)
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 2e-5 * max(1.0, float(np.abs(g["out"]).max()))
    for k in g.files:
        if k.startswith("after/"):
            assert float((sd[k[6:]] - torch.from_numpy(g[k])).abs().max()) <= 1e-6 * max(1.0, float(np.abs(g[k]).max())), k


def _relerr(a, b):
    return float((a.double() - b.double()).norm() / max(1e-12, float(b.double().norm())))


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False], ids=["exact_fp32", "split"])
def test_native_module_matches_reference_fixture(golden_dir, exact):
    from src import ops
    from src.models.transformer import Transformer
    g, sd = _load(golden_dir)
    m = Transformer(**CFG)
    m.load_state_dict(sd, strict=True)
    m.encoder.noise.std = 0.0
    m.cuda().train()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    ops.set_exact_fp32(exact)
    try:
        out = m(x)
        out.backward(torch.from_numpy(g["dout"]).cuda())
        torch.cuda.synchronize()
    finally:
        ops.set_exact_fp32(False)
    assert float((out.detach().cpu() - torch.from_numpy(g["out"])).abs().max()) <= 1e-3 * max(1.0, float(np.abs(g["out"]).max()))
    assert _relerr(x.grad.cpu(), torch.from_numpy(g["dx"])) < 3e-3
    gmax = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("grad/"))
    for k, p in m.named_parameters():
        ref = torch.from_numpy(g["grad/" + k])
        if k in ZERO:
            assert float(ref.abs().max()) < 1e-3 * gmax and float(p.grad.abs().max()) < 1e-3 * gmax, k
            continue
        assert _relerr(p.grad.cpu(), ref) < 3e-3, (k, _relerr(p.grad.cpu(), ref))
    after = m.state_dict()
    for k in g.files:
        if k.startswith("after/"):
            assert float((after[k[6:]].cpu() - torch.from_numpy(g[k])).abs().max()) <= 1e-4 * max(1.0, float(np.abs(g[k]).max())), k
    m.eval()
    with torch.no_grad():
        assert tuple(m(x).shape) == (8, 2) and tuple(m.encode(x).shape) == (8, 64)


@pytest.mark.gpu
def test_attention_layernorm_gelu_ops_match_torch():
    torch.manual_seed(5)
    for S, B, D, H, masked in ((21, 3, 32, 4, True), (197, 2, 48, 3, False)):     # the 0D encoder's sequence; ViViT's 196+1 tokens
        _check_attention(S, B, D, H, masked, True, False)
    # no mask, no dropout: the matrix-core kernels (d_head 16 / 32 / 64; ragged last tiles; both layouts)
    for S, B, D, H, bf in ((197, 2, 48, 3, True), (197, 3, 128, 2, True), (50, 3, 64, 2, False), (256, 1, 32, 1, True), (22, 4, 256, 4, True),
                           (16, 2, 16, 1, False)):
        _check_attention(S, B, D, H, False, False, bf)
    _check_ln_gelu()


def _check_attention(S, B, D, H, masked, use_drop, batch_first):
    from src.models._unit import AttentionFunction
    dh = D // H
    qkv = torch.randn(S, B, 3 * D)
    mask = torch.triu(torch.full((S, S), float("-inf")), diagonal=1) if masked else torch.zeros(S, S)
    drop = (torch.rand(B * H, S, S) > 0.2).float() / 0.8 if use_drop else torch.ones(B * H, S, S)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = qr.chunk(3, dim=2)
    hd = lambda t: t.reshape(S, B * H, dh).transpose(0, 1)
    p = torch.softmax(hd(q) @ hd(k).transpose(1, 2) / math.sqrt(dh) + mask, dim=2) * drop
    ref = (p @ hd(v)).transpose(0, 1).reshape(S, B, D)
    dout = torch.randn(S, B, D)
    ref.backward(dout)
    lay = (lambda t: t.transpose(0, 1).contiguous()) if batch_first else (lambda t: t)
    qg = lay(qkv).cuda().requires_grad_(True)
    out = AttentionFunction.apply(qg, mask.cuda() if masked else None, H, drop.cuda() if use_drop else None, batch_first)
    out.backward(lay(dout).cuda())
    tag = (S, B, D, H, masked, use_drop, batch_first)
    assert float((out.detach().cpu() - lay(ref.detach())).abs().max()) < 3e-6, tag
    assert float((qg.grad.cpu() - lay(qr.grad)).abs().max()) < 3e-5, tag
    if not masked and use_drop:
        qg2 = lay(qkv).cuda().requires_grad_(True)
        out2 = AttentionFunction.apply(qg2, torch.zeros(S, S).cuda(), H, drop.cuda(), batch_first); out2.backward(lay(dout).cuda())
        assert torch.equal(out2, out) and torch.equal(qg2.grad, qg.grad)


def _check_ln_gelu():
    from src.models._unit import AddLayerNormFunction, GeluFunction
    # residual add + LayerNorm
    a = torch.randn(40, 96); b = torch.randn(40, 96); ga = torch.rand(96) + 0.5; be = torch.randn(96)
    ar, br, gr, ber = (t.clone().requires_grad_(True) for t in (a, b, ga, be))
    ref = F.layer_norm(ar + br, (96,), gr, ber, 1e-5); d = torch.randn(40, 96); ref.backward(d)
    ag, bg, gg, beg = (t.cuda().requires_grad_(True) for t in (a, b, ga, be))
    o = AddLayerNormFunction.apply(ag, bg, gg, beg, 1e-5); o.backward(d.cuda())
    for x_, y_ in ((o.detach(), ref.detach()), (ag.grad, ar.grad), (bg.grad, br.grad), (gg.grad, gr.grad), (beg.grad, ber.grad)):
        assert float((x_.cpu() - y_).abs().max()) < 2e-5 * max(1.0, float(y_.abs().max()))
    # GELU, both forms
    x = torch.linspace(-6, 6, 1001)
    for kind, fn in ((0, F.gelu), (1, ot._gelu_tanh)):
        xr = x.clone().requires_grad_(True); yr = fn(xr); yr.backward(torch.ones_like(x))
        xg = x.cuda().requires_grad_(True); yg = GeluFunction.apply(xg, kind); yg.backward(torch.ones_like(xg))
        assert float((yg.detach().cpu() - yr.detach()).abs().max()) < 2e-6 and float((xg.grad.cpu() - xr.grad).abs().max()) < 2e-6

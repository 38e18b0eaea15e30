"""GPU: the whole classifier at input shapes other than the fixtures' and the BASELINE one (batch 1, the reference's
default 112x112 clips, non-square frames, odd frame counts).  The split-arithmetic path (LDS-patch kernels with all their
geometry-dependent forms: pixel-pair stem, residue-class data gradient, eight-wave variants, channel splitting for few
boxes, side-stream schedule) is compared with the exact-fp32 path (gather kernels, one code path for every geometry), which
the fixtures pin to the reference.  Tolerances: logits 1e-3 of their scale; gradients by cosine similarity per
parameter tensor (> 0.9995 for all but at most one tensor per shape: a LeakyReLU kink flip moves one upstream gradient,
see DESIGN.md section 2) and relative L2 over all parameters < 2e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src import ops
    from src.models.R2Plus1D import R2Plus1DClassifier
    from src.loss import FocalLoss

from oracle import r2plus1d as orc

DEV = "cuda:0"
SHAPES = [
    # B, T, H, W, layer_sizes
    (1, 8, 112, 112, [1, 1, 1, 1]),
    (2, 21, 128, 128, [1, 2, 2, 1]),
    (3, 10, 96, 160, [1, 1, 1, 1]),
    (5, 9, 64, 64, [2, 1, 1, 1]),
    (2, 16, 130, 98, [1, 1, 1, 1]),      # odd pooled sizes on the way down
]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"B{s[0]}T{s[1]}_{s[2]}x{s[3]}" for s in SHAPES])
def test_split_path_matches_exact_path(shape):
    B, T, H, W, ls = shape
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = (torch.rand(B, 3, T, H, W, generator=g) * 255 - 100).to(DEV)
    y = (torch.arange(B) % 2).to(DEV)
    params, bufs = orc.synth_state(ls, 5, 0.01)
    sd = dict(params); sd.update(bufs)
    res = {}
    for exact in (True, False):
        ops.set_exact_fp32(exact)
        try:
            model = R2Plus1DClassifier(input_size=(3, T, H, W), num_classes=2, layer_sizes=ls, alpha=0.01)
            model.load_state_dict(sd, strict=True)
            model.to(DEV).train()
            logits = model(x)
            FocalLoss(weight=torch.ones(2), gamma=2.0)(logits, y).backward()
            torch.cuda.synchronize()
            res[exact] = (logits.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
        finally:
            ops.set_exact_fp32(False)
    le, ls_ = res[True][0], res[False][0]
    assert float((le - ls_).abs().max()) <= 1e-3 * max(1.0, float(le.abs().max()))
    ge, gs = res[True][1], res[False][1]
    bad, num, den = [], 0.0, 0.0
    for k in ge:
        a, b = ge[k].double().flatten(), gs[k].double().flatten()
        assert torch.isfinite(b).all(), k
        num += float(((a - b) ** 2).sum()); den += float((a ** 2).sum())
        if float(a.norm()) > 1e-12 and k != "linear.0.bias":      # (that bias has an analytically zero gradient)
            cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
            if cos < 0.9995:
                bad.append((k, cos))
    assert len(bad) <= 1, bad
    assert (num / max(den, 1e-300)) ** 0.5 < 2e-2

"""GPU: the mask-free dropout (md_dropout_ctr, md_bias_gelu_drop_ctr, src/ops.py::counter_dropout) that replaces nn.Dropout's mask
tensors in the ViViT blocks (reference src/models/ViViT.py:31-46,85-91).  The reference's masks come from the framework generator of
whatever device it runs on, so there is no bit pattern to match: what is checked is what dropout has to guarantee -- outputs are
exactly 0 or x / keep, the keep rate, independence across call sites / steps / seeds, that the backward pass regenerates the forward's
decisions, that the fused FeedForward pass equals the mask-tensor pass on the same decisions bit for bit, and reproducibility under
torch.manual_seed, eagerly and replayed from a HIP graph."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _lib():
    from src import _native as N
    return N


def _drop(x, state, tag, keep):
    N = _lib()
    out = torch.empty_like(x)
    N.check(N.lib().md_dropout_ctr(C.c_void_p(x.data_ptr()), C.c_void_p(state.data_ptr()), tag, keep, 1.0 / keep, x.numel(),
                                   C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "md_dropout_ctr")
    return out


@pytest.mark.parametrize("n", [1 << 20, 1000003, 7])
@pytest.mark.parametrize("keep", [0.9, 0.5])
def test_decisions_are_bernoulli_and_a_pure_function_of_state_tag_and_index(n, keep):
    st = torch.tensor([1234567, 5], dtype=torch.int64, device=DEV)
    x = torch.randn(n, device=DEV) + 3.0
    a = _drop(x, st, 0, keep)
    kept = a != 0
    assert torch.equal(a[kept], (x * 1.0 * (1.0 / keep))[kept]) or torch.allclose(a[kept], x[kept] / keep, rtol=1e-6, atol=0)
    assert torch.equal(a[~kept], torch.zeros_like(a[~kept]))
    if n > 1000:
        rate = float(kept.float().mean())
        assert abs(rate - keep) < 5 * np.sqrt(keep * (1 - keep) / n), rate
    assert torch.equal(a, _drop(x, st, 0, keep))                               # same (state, tag): same decisions
    # an unaligned view of the same data (the scalar path) draws the same decisions per index
    y = torch.empty(n + 1, device=DEV)[1:]; y.copy_(x)
    assert torch.equal(a, _drop(y, st, 0, keep))
    if n > 1000:
        for other in (_drop(x, st, 1, keep),                                                              # another call site
                      _drop(x, torch.tensor([1234567, 6], dtype=torch.int64, device=DEV), 0, keep),       # the next step
                      _drop(x, torch.tensor([1234568, 5], dtype=torch.int64, device=DEV), 0, keep)):      # another seed
            both = float(((other != 0) & kept).float().mean())
            assert abs(both - keep * keep) < 6 * np.sqrt(keep * keep * (1 - keep * keep) / n), both        # independent of `a`
        # no structure along the index: lag-1 and lag-4 (one Philox call = 4 elements) agreement at the independent rate
        k = kept.float()
        for lag in (1, 4, 1024):
            agree = float((k[lag:] * k[:-lag]).mean())
            assert abs(agree - keep * keep) < 6 * np.sqrt(keep * keep * (1 - keep * keep) / n), (lag, agree)


def test_autograd_backward_regenerates_the_forward_decisions():
    from src import ops
    from src.models._unit import dropout
    torch.manual_seed(7)
    x = torch.randn(333, 64, device=DEV, requires_grad=True)
    with ops.counter_dropout(x.device, True):
        y1 = dropout(x, 0.3, True)
        y2 = dropout(x, 0.3, True)                                      # second call site: other decisions
    with ops.counter_dropout(x.device, True):                           # (a later forward draws its own key in between)
        y3 = dropout(x, 0.3, True)
    assert not torch.equal(y1 != 0, y2 != 0) and not torch.equal(y1 != 0, y3 != 0)
    g = torch.randn_like(y1)
    (gx,) = torch.autograd.grad(y1, x, g)
    assert torch.equal(gx, torch.where(y1 != 0, g * (1.0 / 0.7), torch.zeros_like(g))) or \
        torch.allclose(gx, torch.where(y1 != 0, g / 0.7, torch.zeros_like(g)), rtol=1e-6, atol=0)
    assert torch.equal(gx != 0, (y1 != 0) & (g != 0))
    # eval mode / p = 0: identity, no launch
    assert dropout(x, 0.3, False) is x and dropout(x, 0.0, True) is x


@pytest.mark.parametrize("kind", [0, 1])
def test_fused_feedforward_pass_equals_the_mask_tensor_pass_on_the_same_decisions(kind):
    from src.models._unit import BiasGeluDropCtrFunction, BiasGeluDropFunction
    rows, Cc, keep = 517, 256, 0.8
    st = torch.tensor([99, 3], dtype=torch.int64, device=DEV)
    x = torch.randn(rows, Cc, device=DEV, requires_grad=True); b = torch.randn(Cc, device=DEV, requires_grad=True)
    mask = (_drop(torch.ones(rows * Cc, device=DEV), st, 4, keep) != 0).float().view(rows, Cc)     # the decisions of (st, tag 4)
    y0 = BiasGeluDropFunction.apply(x, b, mask, 1.0 / keep, kind)
    y1 = BiasGeluDropCtrFunction.apply(x, b, st, 4, keep, kind)
    assert torch.equal(y0, y1)
    g = torch.randn_like(y0)
    gx0, gb0 = torch.autograd.grad(y0, (x, b), g)
    gx1, gb1 = torch.autograd.grad(y1, (x, b), g)
    assert torch.equal(gx0, gx1) and torch.equal(gb0, gb1)


def _vivit(p=0.1):
    from src.models.ViViT import ViViT
    return ViViT(image_size=32, patch_size=8, n_frames=4, n_classes=2, dim=32, depth=1, n_heads=2, pool="mean", in_channels=3, d_head=16,
                 dropout=p, embedd_dropout=p, scale_dim=2).to(DEV).train()


def test_vivit_step_is_reproducible_under_manual_seed_and_draws_new_masks_every_step():
    from src.loss import FocalLoss
    x = torch.randn(2, 3, 4, 32, 32, device=DEV); y = torch.tensor([0, 1], device=DEV)
    lf = FocalLoss(gamma=2.0)

    def run(seed, steps=3):
        torch.manual_seed(seed)
        m = _vivit()
        torch.manual_seed(seed + 1)          # (the model's init consumed the generator: re-seed for the dropout stream itself)
        out = []
        for _ in range(steps):
            for p in m.parameters():
                p.grad = None
            loss = lf(m(x), y); loss.backward()
            out.append((float(loss.detach()), m.mlp[0].weight.grad.clone()))
        return out
    a, b, c = run(11), run(11), run(12)
    for (la, ga), (lb, gb) in zip(a, b):
        assert la == lb and torch.equal(ga, gb)
    assert a[0][0] != a[1][0] and a[1][0] != a[2][0]                    # same weights, same batch: only the masks differ step to step
    assert a[0][0] != c[0][0]
    m = _vivit().eval()
    with torch.no_grad():
        assert torch.equal(m(x), m(x))


def test_vivit_graph_replay_draws_the_masks_of_the_eager_steps():
    """GraphedStep: every replay draws a new key inside the graph (torch's generator offset advances per replay); the warm-up steps
    of the capture do not shift the stream (src/utils/graphed.py::_rng_kept)."""
    from src.loss import FocalLoss
    from src.utils.graphed import GraphedStep
    x = torch.randn(2, 3, 4, 32, 32, device=DEV); y = torch.tensor([0, 1], device=DEV)
    lf = FocalLoss(gamma=2.0)
    torch.manual_seed(5)
    m = _vivit()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    torch.manual_seed(6)
    eager = []
    for _ in range(3):
        for p in m.parameters():
            p.grad = None
        loss = lf(m(x), y); loss.backward()
        eager.append((float(loss.detach()), m.mlp[0].weight.grad.clone()))
    del loss                                   # (a kept loss keeps its autograd graph: GraphedStep refuses to capture next to one)
    m.load_state_dict(sd)
    torch.manual_seed(6)
    gs = GraphedStep(m, lf, [x], y)
    for le, ge in eager:
        _, loss = gs([x], y)
        assert float(loss.detach()) == le and torch.equal(m.mlp[0].weight.grad, ge)


@pytest.mark.parametrize("rows,D", [(517, 128), (33, 64), (100, 192), (7, 32), (1030, 256)])
@pytest.mark.parametrize("p,has_bias", [(0.2, True), (0.0, True), (0.3, False)])
def test_branch_layernorm_pass_matches_the_composed_passes(rows, D, p, has_bias):
    """md_branch_layernorm_* (Linear bias + dropout + residual add + LayerNorm in one pass) against the composed kernels it replaces
    (md_channel_bias_fwd -> md_dropout_ctr -> md_add_layernorm_*) on the SAME dropout decisions: same values up to the order of the
    row sums (1e-5 of the row scale), same bias / gamma / beta gradients (fixed-order column sums of nearly identical inputs)."""
    from src.models._unit import (BranchResidualLayerNormFunction, CtrDropoutFunction, ResidualLayerNormFunction, _ChannelBias)
    torch.manual_seed(rows + D)
    key = torch.empty(2, dtype=torch.int64, device=DEV).random_()
    site = (key, 3) if p > 0 else None
    keep = 1.0 - p

    def leaves():
        g = torch.Generator(device="cpu").manual_seed(1)
        y = torch.randn(rows, D, generator=g).to(DEV).requires_grad_(True)
        b = torch.randn(D, generator=g).to(DEV).requires_grad_(True) if has_bias else None
        r = torch.randn(rows, D, generator=g).to(DEV).requires_grad_(True)
        ga = (torch.rand(D, generator=g) + 0.5).to(DEV).requires_grad_(True)
        be = torch.randn(D, generator=g).to(DEV).requires_grad_(True)
        return y, b, r, ga, be
    y, b, r, ga, be = leaves()
    s1, h1 = BranchResidualLayerNormFunction.apply(y, b, r, ga, be, 1e-5, site, keep)
    y2, b2, r2, ga2, be2 = leaves()
    t = y2 if b2 is None else _ChannelBias.apply(y2[:, :, None], b2)[:, :, 0]
    if site is not None:
        t = CtrDropoutFunction.apply(t, key, 3, keep)
    s2, h2 = ResidualLayerNormFunction.apply(t, r2, ga2, be2, 1e-5)
    assert torch.equal(s1, s2)                                   # the sum itself has no reduction in it: bit-identical
    assert torch.allclose(h1, h2, rtol=1e-5, atol=1e-5)
    gs, gh = torch.randn_like(s1), torch.randn_like(h1)
    torch.autograd.backward([s1, h1], [gs, gh]); torch.autograd.backward([s2, h2], [gs, gh])
    for a, c, name in ((y, y2, "y"), (r, r2, "stream"), (ga, ga2, "gamma"), (be, be2, "beta")) + (((b, b2, "bias"),) if has_bias else ()):
        scale = float(c.grad.abs().max()) + 1e-12
        assert float((a.grad - c.grad).abs().max()) <= 2e-5 * scale, (name, float((a.grad - c.grad).abs().max()), scale)
    if site is not None:                                          # dropped positions carry no gradient to y
        assert torch.equal(y.grad != 0, y2.grad != 0)
    # the stream-only backward (the normalised output unused)
    y, b, r, ga, be = leaves()
    s1, _ = BranchResidualLayerNormFunction.apply(y, b, r, ga, be, 1e-5, site, keep)
    s1.backward(gs)
    assert torch.equal(r.grad, gs)
    expect = gs if site is None else torch.where(y2.grad != 0, gs * (1.0 / keep), torch.zeros_like(gs))
    assert torch.allclose(y.grad, expect, rtol=1e-6, atol=0)

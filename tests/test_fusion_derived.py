"""BASELINE configs 4 and 5 as parity cases (SURVEY 8c "derived fusion oracles"): R2Plus1DClassifier + Transformer-0D with
GradientBlending(Focal), and SlowFast + MLSTM_FCN with GradientBlending(LDAM) - the reference's MultiModalModel_GB recipe applied
to encoder pairs it never wires itself.  The fixture is recorded from the reference's own model classes combined by forward
hooks (tests/golden/make_golden.py::fusion_derived_fixture); weights come from the seeded NumPy recipe.
CPU: the composed oracle restatement against the fixture (logits 5e-5 of their scale, blended loss 1e-5, running statistics).
GPU: the native FusionGB: logits and loss within 1e-3, sub-sampled gradients / gradient norms within 3e-3 (analytically-zero
gradients bounded), running statistics 1e-4."""
import os

import numpy as np
import pytest
import torch

from oracle import fusion as ofu
from oracle import losses as ol

MLSTM = dict(kernel_size=3, stride=1, lstm_n_layers=1, bidirectional=True, alpha=0.01)


def _load(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "fusion_derived.npz"))
    pre = tag + "/shape/"
    shapes = {k[len(pre):]: tuple(int(v) for v in g[k]) for k in g.files if k.startswith(pre)}
    return g, shapes


def _oracle_outs(tag, g, sd):
    xv, xt = torch.from_numpy(g[tag + "/x_vis"]), torch.from_numpy(g[tag + "/x_ts"])
    if tag == "cfg4":
        return ofu.r2p1d_transformer_forward(xv, xt, sd, [1, 1, 1, 1], 0.01, 1, 2, 3, True)
    return ofu.slowfast_mlstm_forward(xv, xt, sd, [1, 1, 1, 1], 4, 1.0, MLSTM, True)


def _oracle_loss(tag, outs, y):
    w = torch.tensor([1.0, 1.0])
    if tag == "cfg4":
        f = lambda o: ol.focal_loss(o, y, w, 2.0)
    else:
        m = ol.ldam_margins([100, 2000], 0.5)
        f = lambda o: ol.ldam_loss(o, y, m, w, 1.0)
    return ol.gradient_blending(f(outs[0]), f(outs[1]), f(outs[2]), 0.1, 0.4, 0.5)


@pytest.mark.parametrize("tag", ["cfg4", "cfg5"])
def test_oracle_matches_derived_fixture(golden_dir, tag):
    g, shapes = _load(golden_dir, tag)
    sd = ofu.fusion_state(shapes, int(g[tag + "/seed"]))
    if tag == "cfg4":                                   # the constant sinusoidal table (transformer.py:10-28)
        from oracle.transformer0d import positional_table
        sd["ts_model.encoder.pos_enc.pe"] = positional_table(*shapes["ts_model.encoder.pos_enc.pe"][::2])
    outs = _oracle_outs(tag, g, sd)
    for i, o in enumerate(outs):
        r = torch.from_numpy(g["%s/out%d" % (tag, i)])
        assert float((o - r).abs().max()) <= 5e-5 * max(1.0, float(r.abs().max())), i
    L = _oracle_loss(tag, outs, torch.from_numpy(g[tag + "/y"]))
    assert abs(float(L) - float(g[tag + "/loss"])) <= 1e-5 * max(1.0, abs(float(g[tag + "/loss"])))
    for k in g.files:
        if k.startswith(tag + "/after/"):
            name = k[len(tag) + 7:]
            assert float((sd[name] - torch.from_numpy(g[k])).abs().max()) <= 2e-6 * max(1.0, float(np.abs(g[k]).max())), name


def _native(tag):
    from src.models.fusion import FusionGB
    from src.models.MLSTM_FCN import MLSTM_FCN
    from src.models.R2Plus1D import R2Plus1DClassifier
    from src.models.slowfast import SlowFast
    from src.models.transformer import Transformer
    if tag == "cfg4":
        vis = R2Plus1DClassifier(input_size=(3, 5, 24, 24), num_classes=2, layer_sizes=[1, 1, 1, 1], alpha=0.01)
        ts = Transformer(n_features=6, kernel_size=3, feature_dims=16, max_len=5, n_layers=1, n_heads=2, dim_feedforward=24, dropout=0.0,
                         cls_dims=12, n_classes=2)
    else:
        vis = SlowFast(input_shape=(3, 8, 32, 32), layers=[1, 1, 1, 1], alpha=4, tau_fast=1, num_classes=2, alpha_elu=1.0)
        ts = MLSTM_FCN(n_features=6, fcn_dim=8, kernel_size=3, stride=1, seq_len=8, lstm_dim=8, lstm_n_layers=1, lstm_bidirectional=True,
                       lstm_dropout=0.0, reduction=4, alpha=0.01, n_classes=2)
    return FusionGB(2, vis, ts)


@pytest.mark.parametrize("tag", ["cfg4", "cfg5"])
def test_native_model_has_the_derived_state_dict(golden_dir, tag):
    g, shapes = _load(golden_dir, tag)
    mine = _native(tag).state_dict()
    assert set(mine) == set(shapes)
    for k, shp in shapes.items():
        assert tuple(mine[k].shape) == shp, k


def _sub(t, n=48):
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // n)][:n]


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["cfg4", "cfg5"])
def test_native_fusion_matches_derived_fixture(golden_dir, tag):
    from src.GradientBlending import GradientBlending
    from src.loss import FocalLoss, LDAMLoss
    g, shapes = _load(golden_dir, tag)
    m = _native(tag)
    missing = m.load_state_dict(ofu.fusion_state(shapes, int(g[tag + "/seed"])), strict=False)
    assert all(k.endswith("pos_enc.pe") for k in missing.missing_keys) and not missing.unexpected_keys
    for mod in m.modules():
        if type(mod).__name__ == "NoiseLayer":
            mod.std = 0.0
    m.cuda().train()
    w = torch.tensor([1.0, 1.0]).cuda()
    loss_fn = FocalLoss(w, 2.0) if tag == "cfg4" else LDAMLoss([100, 2000], max_m=0.5, weight=w, s=1.0)
    gb = GradientBlending(loss_fn, loss_fn, loss_fn, 0.1, 0.4, 0.5)
    outs = m(torch.from_numpy(g[tag + "/x_vis"]).cuda(), torch.from_numpy(g[tag + "/x_ts"]).cuda())
    L = gb(outs[0], outs[1], outs[2], torch.from_numpy(g[tag + "/y"]).cuda())
    L.backward()
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        r = torch.from_numpy(g["%s/out%d" % (tag, i)])
        assert float((o.detach().cpu() - r).abs().max()) <= 1e-3 * max(1.0, float(r.abs().max())), i
    assert abs(float(L) - float(g[tag + "/loss"])) <= 1e-3 * max(1.0, abs(float(g[tag + "/loss"])))
    gmax = max(float(g[k]) for k in g.files if k.startswith(tag + "/gnorm/"))
    for k, p in m.named_parameters():
        rn = float(g["%s/gnorm/%s" % (tag, k)])
        got = p.grad.cpu() if p.grad is not None else torch.zeros(p.shape)
        if rn < 1e-5 * gmax:                              # analytically zero (bias in front of a training-mode BatchNorm, ...)
            assert float(got.double().norm()) < 1e-3 * gmax, k
            continue
        assert abs(float(got.double().norm()) - rn) <= 3e-3 * rn, (k, float(got.double().norm()), rn)
        rs = torch.from_numpy(g["%s/gsub/%s" % (tag, k)])
        assert float((_sub(got) - rs).abs().max()) <= 3e-3 * max(float(rs.abs().max()), rn / max(1.0, got.numel() ** 0.5)), k
    after = m.state_dict()
    for k in g.files:
        if k.startswith(tag + "/after/"):
            name = k[len(tag) + 7:]
            assert float((after[name].cpu() - torch.from_numpy(g[k])).abs().max()) <= 1e-4 * max(1.0, float(np.abs(g[k]).max())), name
    m.update_use_stream("multi")
    assert tuple(m(torch.from_numpy(g[tag + "/x_vis"]).cuda(), torch.from_numpy(g[tag + "/x_ts"]).cuda()).shape) == (4, 2)


@pytest.mark.gpu
def test_train_per_epoch_drives_the_fused_model():
    """The reference's test strategy for its models (test/test_model.py: parameters must change, loss finite) on the cfg4 pair
    through the mirrored train_per_epoch in 'multi-GB' mode with the fused clip + AdamW step."""
    from torch.utils.data import DataLoader, Dataset
    from src.GradientBlending import GradientBlending
    from src.loss import FocalLoss
    from src.optim import ClipAdamW
    from src.train import train_per_epoch, valid_per_epoch

    class Pairs(Dataset):
        def __init__(self):
            g = torch.Generator().manual_seed(5)
            self.v = torch.randn(8, 3, 5, 24, 24, generator=g); self.t = torch.randn(8, 5, 6, generator=g)
            self.y = torch.tensor([0, 1, 0, 1, 1, 0, 0, 1])

        def __len__(self):
            return 8

        def __getitem__(self, i):
            return {"video": self.v[i], "0D": self.t[i]}, self.y[i]

    torch.manual_seed(6)
    m = _native("cfg4").cuda()
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    w = torch.tensor([1.0, 1.0]).cuda()
    gb = GradientBlending(FocalLoss(w, 2.0), FocalLoss(w, 2.0), FocalLoss(w, 2.0), 0.1, 0.4, 0.5)
    opt = ClipAdamW(m.parameters(), lr=1e-3, max_norm=1.0)
    loader = DataLoader(Pairs(), batch_size=4, shuffle=False)
    for _ in range(2):
        loss, acc, f1 = train_per_epoch(loader, m, opt, None, gb, "cuda:0", 1.0, "multi-GB")
    assert np.isfinite(loss) and 0.0 <= acc <= 1.0
    changed = [k for k, v in m.named_parameters() if not torch.equal(v.detach(), before[k])]
    assert len(changed) > 0.9 * len(before)
    vloss, vacc, vf1 = valid_per_epoch(loader, m, opt, gb, "cuda:0", "multi-GB")
    assert np.isfinite(vloss)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["cfg4", "cfg5"])
def test_encoders_on_separate_streams_are_bit_identical_to_one_stream(tag):
    """The 0D encoder beside the video encoder on a side stream (and, for cfg5, the fast pathway beside the slow one;
    src/utils/streams.py) against the same step on one stream: outputs, every parameter gradient and the running statistics
    agree bit for bit, three times in a row."""
    from src.utils import streams
    torch.manual_seed(9)
    m = _native(tag).cuda().train()
    xv = torch.randn(4, 3, 5, 24, 24, device="cuda") if tag == "cfg4" else torch.randn(4, 3, 8, 32, 32, device="cuda")
    xt = torch.randn(4, 5, 6, device="cuda") if tag == "cfg4" else torch.randn(4, 8, 6, device="cuda")
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    old = streams._ENABLED

    def run(flag):
        streams._ENABLED = flag
        m.load_state_dict(sd)
        for p in m.parameters():
            p.grad = None
        torch.manual_seed(10)                      # the 0D Transformer's NoiseLayer draws from the CPU generator
        outs = m(xv, xt)
        sum(o.square().sum() for o in outs).backward()
        torch.cuda.synchronize()
        return ([o.detach().clone() for o in outs], {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None},
                {k: v.clone() for k, v in m.state_dict().items() if "running" in k})
    try:
        ref = run(False)
        for _ in range(3):
            got = run(True)
            for a, b in zip(got[0], ref[0]):
                assert torch.equal(a, b)
            assert got[1].keys() == ref[1].keys()
            for k in ref[1]:
                assert torch.equal(got[1][k], ref[1][k]), k
            for k in ref[2]:
                assert torch.equal(got[2][k], ref[2][k]), k
    finally:
        streams._ENABLED = old


@pytest.mark.gpu
def test_graphed_0d_branch_matches_the_eager_step():
    """MD_GRAPH_BRANCH=1 (src/models/fusion.py): forward and backward of the 0D encoder + head replayed from HIP graphs inside an
    otherwise eager step (R(2+1)D trunk + Transformer-0D, no dropout, NoiseLayer off so that the capture's warm-up steps do not shift
    the CPU generator): outputs and every parameter gradient bit-identical to the eager step, on three successive batches."""
    import src.models.fusion as fu
    torch.manual_seed(21)
    m = _native("cfg4").cuda().train()
    for mod in m.modules():
        if type(mod).__name__ == "NoiseLayer":
            mod.std = 0.0
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    batches = [(torch.randn(4, 3, 5, 24, 24, device="cuda"), torch.randn(4, 5, 6, device="cuda")) for _ in range(3)]

    def run(flag):
        old = fu._GRAPH_BRANCH
        fu._GRAPH_BRANCH = flag
        m.__dict__.pop("_md_ts_graph", None)
        m.load_state_dict(sd)
        res = []
        try:
            for xv, xt in batches:
                for p in m.parameters():
                    p.grad = None
                outs = m(xv, xt)
                sum(o.square().sum() for o in outs).backward()
                torch.cuda.synchronize()
                res.append(([o.detach().clone() for o in outs], {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
            used = m.__dict__.get("_md_ts_graph")
            state = {k: v.detach().clone() for k, v in m.state_dict().items()}
        finally:
            fu._GRAPH_BRANCH = old
            m.__dict__.pop("_md_ts_graph", None)
        return res, used, state

    ref, used0, st0 = run(False)
    import warnings
    warn_always = torch.is_warn_always_enabled()
    torch.set_warn_always(True)
    try:
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            got, used1, st1 = run(True)
    finally:
        torch.set_warn_always(warn_always)
    assert used0 is None and used1 not in (None, False)
    # the replayed steps feed the parameters' AccumulateGrad nodes on the stream those nodes belong to: no cross-stream hand-over per
    # step (PyTorch says so with a warning when there is one)
    assert not [w for w in caught if "AccumulateGrad node's stream does not match" in str(w.message)], [str(w.message)[:120] for w in caught]
    # buffers too: the probe and the warm-up iterations of the capture must not leave extra BatchNorm momentum updates behind, and
    # the replays must advance num_batches_tracked exactly as the eager steps do (graphed.py)
    assert st0.keys() == st1.keys()
    for k in st0:
        assert torch.equal(st0[k], st1[k]), k
    assert any(k.endswith("num_batches_tracked") and int(v) == 3 for k, v in st1.items() if k.startswith("ts_model."))
    for (o0, g0), (o1, g1) in zip(ref, got):
        for a, b in zip(o0, o1):
            assert torch.equal(a, b)
        assert g0.keys() == g1.keys()
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k


@pytest.mark.gpu
def test_graphed_0d_branch_refuses_a_stale_autograd_graph():
    """An eager step on the default stream whose loss tensor is kept alive, then MD_GRAPH_BRANCH: the capture would have to
    synchronise with the default stream (ROCm 7.2 crashes in hipStreamEndCapture); GraphedBranch notices the stale graph in its eager
    probe, FusionGB says so and stays eager -- the step still runs."""
    import src.models.fusion as fu
    torch.manual_seed(22)
    m = _native("cfg4").cuda().train()
    xv, xt = torch.randn(4, 3, 5, 24, 24, device="cuda"), torch.randn(4, 5, 6, device="cuda")
    kept = sum(o.square().sum() for o in m(xv, xt))          # autograd graph alive, built on the default stream
    kept.backward(retain_graph=True)
    old = fu._GRAPH_BRANCH
    fu._GRAPH_BRANCH = True
    try:
        for p in m.parameters():
            p.grad = None
        outs = m(xv, xt)
        assert m.__dict__.get("_md_ts_graph") is False         # refused, not crashed
        sum(o.square().sum() for o in outs).backward()
        assert all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None)
    finally:
        fu._GRAPH_BRANCH = old
        m.__dict__.pop("_md_ts_graph", None)
    del kept

"""Scope row a16: MultiModalModel / MultiModalModel_GB / TFN / TFN_GB (src/models/MultiModal.py:10-331).
CPU: the oracle restatement against the fixture recorded from the reference (outputs 2e-5, running statistics 1e-6); the mirror's
state-dict keys equal the reference's.  GPU: the native modules against the same fixture: every output within 1e-3 of its scale,
parameter gradients within 3e-3 relative L2 (gradients that are analytically zero - biases in front of a training-mode BatchNorm -
are bounded on both sides), running statistics 1e-4; the tensor-fusion kernel against torch.bmm on the CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import multimodal as om

AV = dict(image_size=32, patch_size=8, n_frames=5, dim=16, depth=1, n_heads=2, in_channels=3, d_head=8, dropout=0.0,
          embedd_dropout=0.0, scale_dim=2)
A0 = dict(n_features=6, kernel_size=3, feature_dims=16, max_len=5, n_layers=1, n_heads=2, dim_feedforward=24, dropout=0.0)
AVG = dict(AV, n_classes=2, pool="cls", alpha=1.0)
A0G = dict(A0, cls_dims=12, n_classes=2)
CASES = {"mm": ("MultiModalModel", dict(AV, pool="mean"), A0, om.multimodal_forward),
         "gb": ("MultiModalModel_GB", AVG, A0G, om.multimodal_gb_forward),
         "tfn": ("TFN", dict(AV, pool="mean"), A0, om.tfn_forward),
         "tfngb": ("TFN_GB", AVG, A0G, om.tfn_gb_forward)}


def _load(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "multimodal.npz"))
    pre = tag + "/sd/"
    return g, {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}


def _outs(g, tag, what="out"):
    return [torch.from_numpy(g[k]) for k in sorted(k for k in g.files if k.startswith("%s/%s" % (tag, what)))]


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_matches_reference_fixture(golden_dir, tag):
    g, sd = _load(golden_dir, tag)
    sd = {k: v.clone() for k, v in sd.items()}
    outs = CASES[tag][3](torch.from_numpy(g[tag + "/x_vis"]), torch.from_numpy(g[tag + "/x_ts"]), sd)
    outs = outs if isinstance(outs, tuple) else (outs,)
    refs = _outs(g, tag)
    assert len(outs) == len(refs)
    for o, r in zip(outs, refs):
        assert float((o - r).abs().max()) <= 2e-5 * max(1.0, float(r.abs().max()))
    for k in g.files:
        if k.startswith(tag + "/after/"):
            name = k[len(tag) + 7:]
            assert float((sd[name] - torch.from_numpy(g[k])).abs().max()) <= 1e-6 * max(1.0, float(np.abs(g[k]).max())), name


@pytest.mark.parametrize("tag", list(CASES))
def test_mirror_has_the_reference_state_dict(golden_dir, tag):
    import importlib
    MM = importlib.import_module("src.models.MultiModal")
    g, sd = _load(golden_dir, tag)
    name, av, a0, _ = CASES[tag]
    m = getattr(MM, name)(2, dict(av), dict(a0))
    mine = m.state_dict()
    assert set(mine) == set(sd)
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k


def _relerr(a, b):
    return float((a.double() - b.double()).norm() / max(1e-12, float(b.double().norm())))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_native_module_matches_reference_fixture(golden_dir, tag):
    from src.models import MultiModal as MM
    g, sd = _load(golden_dir, tag)
    name, av, a0, _ = CASES[tag]
    m = getattr(MM, name)(2, dict(av), dict(a0))
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if type(mod).__name__ == "NoiseLayer":
            mod.std = 0.0
    m.cuda().train()
    outs = m(torch.from_numpy(g[tag + "/x_vis"]).cuda(), torch.from_numpy(g[tag + "/x_ts"]).cuda())
    outs = outs if isinstance(outs, tuple) else (outs,)
    refs, douts = _outs(g, tag), _outs(g, tag, "dout")
    assert len(outs) == len(refs)
    sum((o * d.cuda()).sum() for o, d in zip(outs, douts)).backward()
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert float((o.detach().cpu() - r).abs().max()) <= 1e-3 * max(1.0, float(r.abs().max()))
    gmax = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith(tag + "/grad/"))
    for k, p in m.named_parameters():
        r = torch.from_numpy(g["%s/grad/%s" % (tag, k)])
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(r)
        if float(r.abs().max()) < 1e-4 * gmax:           # analytically zero in the reference (up to its rounding)
            assert float(got.abs().max()) < 1e-3 * gmax, k
            continue
        assert _relerr(got, r) < 3e-3, (k, _relerr(got, r))
    after = m.state_dict()
    for k in g.files:
        if k.startswith(tag + "/after/"):
            name_ = k[len(tag) + 7:]
            assert float((after[name_].cpu() - torch.from_numpy(g[k])).abs().max()) <= 1e-4 * max(1.0, float(np.abs(g[k]).max())), name_
    if tag in ("gb", "tfngb"):
        lat = m.vis_latent if tag == "gb" else m.h_vis
        assert isinstance(lat, tuple) and tuple(lat[0].shape) == (4, 16)
    m.eval()
    enc = m.encode(torch.from_numpy(g[tag + "/x_vis"]).cuda(), torch.from_numpy(g[tag + "/x_ts"]).cuda())
    assert len(enc) == 3 and enc[1].shape == (4, 16) and enc[2].shape == (4, 16)


@pytest.mark.gpu
def test_gb_stream_switch():
    from src.models.MultiModal import MultiModalModel_GB
    torch.manual_seed(3)
    m = MultiModalModel_GB(2, dict(AVG), dict(A0G), use_stream="video").cuda()
    xv = torch.randn(2, 3, 5, 32, 32, device="cuda"); xt = torch.randn(2, 5, 6, device="cuda")
    assert tuple(m(xv, xt).shape) == (2, 2)
    m.update_use_stream("0D"); assert tuple(m(xv, xt).shape) == (2, 2)
    m.update_use_stream("multi"); assert tuple(m(xv, xt).shape) == (2, 2)
    m.update_use_stream("multi-GB"); out = m(xv, xt)
    assert isinstance(out, tuple) and len(out) == 3
    m.remove_my_hooks()


@pytest.mark.gpu
def test_outer_fusion_matches_bmm():
    from src.models._unit import OuterFusionFunction
    torch.manual_seed(4)
    a = torch.randn(5, 37); c = torch.randn(5, 21); d = torch.randn(5, 38 * 22)
    ar, cr = a.clone().requires_grad_(True), c.clone().requires_grad_(True)
    one = torch.ones(5, 1)
    ref = torch.bmm(torch.cat((one, ar), 1).unsqueeze(2), torch.cat((one, cr), 1).unsqueeze(1)).view(5, -1)
    ref.backward(d)
    ag, cg = a.cuda().requires_grad_(True), c.cuda().requires_grad_(True)
    out = OuterFusionFunction.apply(ag, cg); out.backward(d.cuda())
    assert torch.equal(out.detach().cpu(), ref.detach())
    assert float((ag.grad.cpu() - ar.grad).abs().max()) < 2e-5 and float((cg.grad.cpu() - cr.grad).abs().max()) < 2e-5

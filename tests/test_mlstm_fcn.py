"""Scope row a13: MLSTM_FCN (src/models/MLSTM_FCN.py:16-169; BASELINE configs[0] is this model's plumbing).
CPU: the oracle restatement against the fixture recorded from the reference (logits 1e-5, running statistics 1e-6).
GPU: the native module against the same fixture: logits within 1e-3 of their scale, input and parameter gradients within 3e-3
relative L2 (BatchNorm over 8 samples; both arithmetic modes), running statistics 1e-4; parameters whose gradient is
analytically zero (attention weights; biases in front of a BatchNorm) are bounded on both sides instead.  Also: LSTM
inter-layer dropout (training mode) keeps the scale of the activations and is switched off in eval mode."""
import os

import numpy as np
import pytest
import torch

from oracle import mlstm_fcn as om

CFG = dict(n_features=14, fcn_dim=32, kernel_size=3, stride=1, seq_len=21, lstm_dim=24, lstm_n_layers=2, lstm_bidirectional=True,
           lstm_dropout=0.0, reduction=16, alpha=0.01, n_classes=2)
ZERO = ("rnn.w_s1.weight", "rnn.w_s1.bias", "rnn.w_s2.weight", "rnn.w_s2.bias", "fcn.0.conv.bias", "fcn.2.conv.bias",
        "classifier.0.bias", "converter.bias")     # (converter.bias: a constant shift in front of Linear + BatchNorm)


def _load(golden_dir):
    g = np.load(os.path.join(golden_dir, "mlstm_fcn.npz"))
    return g, {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}


def test_oracle_matches_reference_fixture(golden_dir):
    g, sd = _load(golden_dir)
    sd = {k: v.clone() for k, v in sd.items()}
    out = om.mlstm_fcn_forward(torch.from_numpy(g["x"]), sd, 3, 1, 2, True, 0.01, True)
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 1e-5 * max(1.0, float(np.abs(g["out"]).max()))
    for k in g.files:
        if k.startswith("after/"):
            assert float((sd[k[6:]] - torch.from_numpy(g[k])).abs().max()) <= 1e-6 * max(1.0, float(np.abs(g[k]).max())), k


def _relerr(a, b):
    return float((a.double() - b.double()).norm() / max(1e-12, float(b.double().norm())))


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False], ids=["exact_fp32", "split"])
def test_native_module_matches_reference_fixture(golden_dir, exact):
    from src import ops
    from src.models.MLSTM_FCN import MLSTM_FCN
    g, sd = _load(golden_dir)
    m = MLSTM_FCN(**CFG)
    m.load_state_dict(sd, strict=True)
    m.noise.std = 0.0
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
        assert tuple(m(x).shape) == (8, 2) and tuple(m.encode(x).shape) == (8, 2 * 24 + 2 * 32)


@pytest.mark.gpu
def test_lstm_interlayer_dropout():
    from src.models._unit import lstm_forward
    torch.manual_seed(3)
    lstm = torch.nn.LSTM(6, 16, num_layers=3, bidirectional=True, dropout=0.5).cuda()
    x = torch.randn(9, 64, 6, device="cuda")
    lstm.eval()
    a = lstm_forward(x, lstm); b = lstm_forward(x, lstm)
    assert torch.equal(a, b)                                        # eval: no dropout, deterministic
    lstm.train()
    c = lstm_forward(x, lstm); d = lstm_forward(x, lstm)
    assert not torch.equal(c, d)                                    # training: a fresh mask per call
    assert 0.5 < float(c.abs().mean() / a.abs().mean()) < 2.0       # inverted dropout keeps the scale


@pytest.mark.gpu
def test_cfg1_batch32_head_is_composed_from_units():
    """BASELINE configs[0] (script defaults, batch 32): the 512 -> 256 head at B = 32 does not fit the fused head kernels' LDS tiles
    and is composed from the Linear+BatchNorm1d unit and the MFMA Linear (models/_unit.py::head_apply); logits, input gradient
    and parameter gradients against the oracle restatement (pinned by the reference fixture above) run on the CPU."""
    from oracle import mlstm_fcn as om
    from src.models.MLSTM_FCN import MLSTM_FCN
    torch.manual_seed(21)
    m = MLSTM_FCN(n_features=14, fcn_dim=128, kernel_size=3, stride=1, seq_len=21, lstm_dim=128, lstm_n_layers=4, lstm_bidirectional=True,
                  lstm_dropout=0.0, reduction=16, alpha=0.01, n_classes=2)
    m.noise.std = 0.0
    assert 2 * 32 * m.classifier[0].out_features * 4 > 60000
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    ref_sd = dict(sd); ref_sd.update(leaves)
    x = torch.randn(32, 21, 14); dout = torch.randn(32, 2)
    xr = x.clone().requires_grad_(True)
    ref = om.mlstm_fcn_forward(xr, ref_sd, kernel_size=3, stride=1, lstm_n_layers=4, bidirectional=True, alpha=0.01, training=True)
    ref.backward(dout)
    m.cuda().train()
    xg = x.cuda().requires_grad_(True)
    out = m(xg)
    out.backward(dout.cuda())
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / max(1e-12, float(b.double().norm())))
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 1e-3 * max(1.0, float(ref.abs().max()))
    assert rel(xg.grad, xr.grad) < 3e-3
    gmax = max(float(v.grad.abs().max()) for v in leaves.values() if v.grad is not None)
    for k, p in m.named_parameters():
        r = leaves[k].grad
        if r is None or float(r.abs().max()) < 1e-4 * gmax:
            assert p.grad is None or float(p.grad.abs().max()) < 1e-3 * gmax, k
            continue
        assert rel(p.grad, r) < 3e-3, (k, rel(p.grad, r))
    # eval mode through the composed head (running statistics, folded Linear bias)
    m.eval()
    sd_eval = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        got = m(x.cuda()).cpu()
        ref_eval = om.mlstm_fcn_forward(x, sd_eval, kernel_size=3, stride=1, lstm_n_layers=4, bidirectional=True, alpha=0.01, training=False)
    assert float((got - ref_eval).abs().max()) <= 1e-3 * max(1.0, float(ref_eval.abs().max()))

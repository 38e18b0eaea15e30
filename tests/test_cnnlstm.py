"""Scope row a14: CnnLSTM (src/models/CnnLSTM.py:10-109).
CPU: the oracle restatement against the fixture recorded from the reference (logits 1e-5, running statistics 1e-6).
GPU: the native module against the same fixture: logits within 1e-3 of their scale (measured ~1e-5), input and parameter
gradients within 2e-3 relative L2 (BatchNorm over 6 samples; both arithmetic modes), running statistics 1e-4.  The attention
parameters w_s1 / w_s2 have analytically zero gradients (see the mirror's docstring): the reference's are round-off noise
(< 1e-9), the mirror's are exactly zero."""
import os

import numpy as np
import pytest
import torch

from oracle import cnnlstm as oc

CFG = dict(seq_len=21, n_features=12, conv_dim=32, conv_kernel=3, conv_stride=1, conv_padding=1, lstm_dim=32, n_layers=2,
           bidirectional=True, n_classes=2)


def _load(golden_dir):
    g = np.load(os.path.join(golden_dir, "cnnlstm.npz"))
    return g, {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}


def test_oracle_matches_reference_fixture(golden_dir):
    g, sd = _load(golden_dir)
    sd = {k: v.clone() for k, v in sd.items()}
    out = oc.cnnlstm_forward(torch.from_numpy(g["x"]), sd, CFG["lstm_dim"], CFG["n_layers"], CFG["bidirectional"], True)
    assert float((out - torch.from_numpy(g["out"])).abs().max()) <= 1e-5 * max(1.0, float(np.abs(g["out"]).max()))
    for k in g.files:
        if k.startswith("after/"):
            assert float((sd[k[6:]] - torch.from_numpy(g[k])).abs().max()) <= 1e-6 * max(1.0, float(np.abs(g[k]).max())), k
    for k in ("w_s1.weight", "w_s2.weight"):
        assert float(np.abs(g["grad/" + k]).max()) < 1e-8          # the reference's own attention gradients are noise


def _relerr(a, b):
    return float((a.double() - b.double()).norm() / max(1e-12, float(b.double().norm())))


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False], ids=["exact_fp32", "split"])
def test_native_module_matches_reference_fixture(golden_dir, exact):
    from src import ops
    from src.models.CnnLSTM import CnnLSTM
    g, sd = _load(golden_dir)
    m = CnnLSTM(**CFG)
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
    assert _relerr(x.grad.cpu(), torch.from_numpy(g["dx"])) < 2e-3
    gmax = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("grad/"))
    for k, p in m.named_parameters():
        ref = torch.from_numpy(g["grad/" + k])
        if k.startswith("w_s") or k in ("conv.1.bias", "classifier.0.bias"):     # analytically zero: attention weights, biases
            assert float(ref.abs().max()) < 1e-3 * gmax and float(p.grad.abs().max()) < 1e-3 * gmax, k   # in front of a BatchNorm
            continue
        assert _relerr(p.grad.cpu(), ref) < 2e-3, (k, _relerr(p.grad.cpu(), ref))
    after = m.state_dict()
    for k in g.files:
        if k.startswith("after/"):
            assert float((after[k[6:]].cpu() - torch.from_numpy(g[k])).abs().max()) <= 1e-4 * max(1.0, float(np.abs(g[k]).max())), k
    m.eval()
    with torch.no_grad():
        assert tuple(m(x).shape) == (6, 2) and tuple(m.encode(x).shape) == (6, 64)

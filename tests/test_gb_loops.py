"""Scope row (f)-1: GB_estimate / train_GB_dynamic (src/GradientBlending.py:52-114, 310-446) -- host logic, CPU.
The mirrored loops, driven by the same tiny two-stream model and loaders as the reference was when the fixture was recorded
(oracle/fake_multimodal.py), must reproduce the reference's estimated weights, loss histories (1e-5: float32 device-side sums
here vs Python float sums there), accuracies / F1 (exact) and final blending weights (1e-4), including its literal quirks;
literal=False gives the corrected variant (different numbers, same invariants)."""
import os

import numpy as np
import pytest
import torch

from oracle.fake_multimodal import FakeMultiModalGB, loaders
from src.GradientBlending import GB_estimate, GradientBlending, train_GB_dynamic


def _ce():
    return torch.nn.CrossEntropyLoss(reduction="sum")


def test_gb_estimate_matches_reference(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "gb_loops.npz"))
    tr, va = loaders(11)
    model = FakeMultiModalGB()
    last = str(tmp_path / "last.pt")
    torch.save(model.state_dict(), last)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    w = GB_estimate(2, tr, va, last, model, opt, None, _ce(), "cpu", None)
    assert set(w) == {"video", "0D", "multi"} and abs(sum(w.values()) - 1.0) < 1e-12
    for k in w:
        assert abs(w[k] - float(g["est/" + k])) <= 1e-4 * max(1.0, abs(float(g["est/" + k]))), (k, w[k], float(g["est/" + k]))
    # corrected variant: lists reset per task -> other numbers, still normalised
    model = FakeMultiModalGB(); torch.save(model.state_dict(), last)
    w2 = GB_estimate(2, tr, va, last, model, torch.optim.SGD(model.parameters(), lr=0.05), None, _ce(), "cpu", None, literal=False)
    assert abs(sum(w2.values()) - 1.0) < 1e-12 and any(abs(w2[k] - w[k]) > 1e-6 for k in w)


def test_train_gb_dynamic_matches_reference(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "gb_loops.npz"))
    tr, va = loaders(11)
    model = FakeMultiModalGB()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    loss_gb = GradientBlending(_ce(), _ce(), _ce(), 0.2, 0.3, 0.5, 1.0)
    hist = train_GB_dynamic(tr, va, model, opt, None, loss_gb, _ce(), "cpu", num_epoch=4, epoch_per_GB_estimate=2,
                            num_epoch_GB_estimate=2, verbose=None, save_best_dir=str(tmp_path / "best.pt"),
                            save_last_dir=str(tmp_path / "last.pt"), exp_dir=str(tmp_path / "exp"), max_norm_grad=1.0,
                            criteria="loss")
    for name, h in zip(("train_loss", "train_acc", "train_f1", "valid_loss", "valid_acc", "valid_f1"), hist):
        ref = g["dyn/" + name]
        assert len(h) == len(ref)
        tol = 1e-5 if "loss" in name else 1e-12
        assert np.allclose(np.array(h, dtype=np.float64), ref, rtol=tol, atol=tol), (name, h, ref)
    got = np.array([loss_gb.vis_weight, loss_gb.ts_weight, loss_gb.vis_ts_weight], dtype=np.float64)
    assert np.allclose(got, g["dyn/weights"], rtol=1e-4, atol=1e-6), (got, g["dyn/weights"])
    assert os.path.isfile(tmp_path / "best.pt") and os.path.isfile(tmp_path / "last.pt")

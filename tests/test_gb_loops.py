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
from src.GradientBlending import GB_estimate, GradientBlending, evaluate_GB, train_GB, train_GB_dynamic


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


def test_train_gb_and_evaluate_gb_match_reference(golden_dir, tmp_path):
    """train_GB (fixed weights, criteria "acc": GradientBlending.py:165-308) and evaluate_GB (:116-163, macro-F1 of the fused /
    vision / 0D heads) against the reference's own run on the same tiny model: histories, the three per-stream scores over
    both loaders, and WHICH epoch was kept as best.pt (the head bias of the saved checkpoint)."""
    g = np.load(os.path.join(golden_dir, "gb_loops.npz"))
    tr, va = loaders(11)
    model = FakeMultiModalGB(); model.update_use_stream("multi-GB")
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    loss_gb = GradientBlending(_ce(), _ce(), _ce(), 0.2, 0.3, 0.5, 1.0)
    best = str(tmp_path / "best.pt")
    hist = train_GB(tr, va, model, opt, None, loss_gb, "cpu", num_epoch=3, verbose=None, save_best_dir=best,
                    save_last_dir=str(tmp_path / "last.pt"), exp_dir=str(tmp_path / "exp2"), max_norm_grad=1.0, criteria="acc")
    for name, h in zip(("train_loss", "train_acc", "train_f1", "valid_loss", "valid_acc", "valid_f1"), hist):
        tol = 1e-5 if "loss" in name else 1e-12
        assert np.allclose(np.array(h, dtype=np.float64), g["fix/" + name], rtol=tol, atol=tol), (name, h, g["fix/" + name])
    assert np.allclose(evaluate_GB(tr, model, opt, "cpu", 0.5), g["fix/evalgb_train"], atol=1e-12)
    assert np.allclose(evaluate_GB(va, model, opt, "cpu", 0.5), g["fix/evalgb_valid"], atol=1e-12)
    # the per-epoch monitor of train_GB: its last entry is what evaluate_GB gives on the final model
    assert np.allclose(train_GB.stream_f1["valid"][-1], g["fix/evalgb_valid"], atol=1e-12) and len(train_GB.stream_f1["train"]) == 3
    sd = torch.load(best, weights_only=True)
    assert np.allclose(sd["head.bias"].double().numpy(), g["fix/best_head_bias"], rtol=1e-5, atol=1e-7)


# ---- the same loops on the GPU, with the native MultiModalModel_GB ------------------------------------------------------------
AV = dict(image_size=32, patch_size=8, n_frames=5, dim=16, depth=1, n_heads=2, in_channels=3, d_head=8, dropout=0.0,
          embedd_dropout=0.0, scale_dim=2, n_classes=2, pool="cls", alpha=1.0)
A0 = dict(n_features=6, kernel_size=3, feature_dims=16, max_len=5, n_layers=1, n_heads=2, dim_feedforward=24, dropout=0.0,
          cls_dims=12, n_classes=2)


def _native_and_oracle(golden_dir):
    from oracle.fake_multimodal import OracleMultiModalGB
    from src.models.MultiModal import MultiModalModel_GB
    g = np.load(os.path.join(golden_dir, "multimodal.npz"))
    sd = {k[len("gb/sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("gb/sd/")}
    m = MultiModalModel_GB(2, dict(AV), dict(A0))
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if type(mod).__name__ == "NoiseLayer":
            mod.std = 0.0
    ref = OracleMultiModalGB(sd, {k for k, _ in m.named_parameters()})
    return m.cuda(), ref


def _close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.allclose(a, b, rtol=tol, atol=tol)


@pytest.mark.gpu
def test_gb_estimate_native_model_on_gpu_tracks_the_oracle_model(golden_dir, tmp_path):
    """GB_estimate (checkpoint reload before each task, update_use_stream("video" / "0D" / "multi"), 2 epochs each) driven on
    cuda:0 with the native ViViT + 0D-Transformer fusion model, against the same loop driven on the CPU with the fp64 oracle
    model from the same initial weights: the per-task loss curves agree to 1e-3, hence the blending weights (an
    ill-conditioned ratio of loss differences) to 2e-2, and the reloaded checkpoint is really what each task starts from."""
    from oracle.fake_multimodal import clip_loaders
    tr, va = clip_loaders()
    m, ref = _native_and_oracle(golden_dir)
    seen = {}

    def run(model, dev, tag):
        last = str(tmp_path / (tag + "_last.pt"))
        torch.save(model.state_dict(), last)
        opt = torch.optim.SGD(model.parameters(), lr=0.02)
        curves = []
        import src.GradientBlending as GB
        orig_t, orig_v = GB.train_per_epoch, GB.valid_per_epoch

        def spy_t(*a, **k):
            if tag == "hip":            # first batch of each task must see the reloaded weights: record a parameter checksum
                seen.setdefault(model.use_stream, float(sum(p.detach().double().sum() for p in model.parameters())))
            r = orig_t(*a, **k); curves.append(("t", model.use_stream, r[0])); return r

        def spy_v(*a, **k):
            r = orig_v(*a, **k); curves.append(("v", model.use_stream, r[0])); return r
        GB.train_per_epoch, GB.valid_per_epoch = spy_t, spy_v
        try:
            w = GB_estimate(2, tr, va, last, model, opt, None, _ce(), dev, 1.0)
        finally:
            GB.train_per_epoch, GB.valid_per_epoch = orig_t, orig_v
        return w, curves

    w_ref, c_ref = run(ref, "cpu", "ref")
    start = float(sum(p.detach().double().sum() for p in m.parameters()))
    w_hip, c_hip = run(m, "cuda:0", "hip")
    assert [c[:2] for c in c_hip] == [c[:2] for c in c_ref]
    assert [c[1] for c in c_hip[::4]] == ["video", "0D", "multi"]
    for (k, s, a), (_, _, b) in zip(c_hip, c_ref):
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (k, s, a, b)
    for task in ("video", "0D", "multi"):           # every task started from the saved checkpoint, not from the previous task's end
        assert abs(seen[task] - start) <= 1e-4 * max(1.0, abs(start)), (task, seen[task], start)
    assert abs(sum(w_hip.values()) - 1.0) < 1e-9 and all(np.isfinite(v) for v in w_hip.values())
    for k in w_ref:
        assert abs(w_hip[k] - w_ref[k]) <= 2e-2 * max(1.0, abs(w_ref[k])), (k, w_hip[k], w_ref[k])
    assert all(p.is_cuda for p in m.parameters())


@pytest.mark.gpu
def test_train_gb_dynamic_native_model_on_gpu_tracks_the_oracle_model(golden_dir, tmp_path):
    """train_GB_dynamic (3 epochs, re-estimation due on epochs 1 and 2 under the reference's literal schedule test) on cuda:0 with
    the native model against the fp64 oracle model on the CPU: loss histories 2e-3, accuracies identical, final blending
    weights 5e-2 absolute (they are normalised to 1), model left in "multi-GB" mode, best/last checkpoints loadable."""
    from oracle.fake_multimodal import clip_loaders
    tr, va = clip_loaders()
    m, ref = _native_and_oracle(golden_dir)
    out = {}
    for tag, model, dev in (("ref", ref, "cpu"), ("hip", m, "cuda:0")):
        opt = torch.optim.SGD(model.parameters(), lr=0.02)
        loss_gb = GradientBlending(_ce(), _ce(), _ce(), 0.2, 0.3, 0.5, 1.0)
        hist = train_GB_dynamic(tr, va, model, opt, None, loss_gb, _ce(), dev, num_epoch=3, epoch_per_GB_estimate=2,
                                num_epoch_GB_estimate=2, verbose=None, save_best_dir=str(tmp_path / (tag + "_best.pt")),
                                save_last_dir=str(tmp_path / (tag + "_last.pt")), exp_dir=str(tmp_path / (tag + "_exp")),
                                max_norm_grad=1.0, criteria="loss")
        out[tag] = (hist, np.array([loss_gb.vis_weight, loss_gb.ts_weight, loss_gb.vis_ts_weight], dtype=np.float64))
    (h_hip, w_hip), (h_ref, w_ref) = out["hip"], out["ref"]
    for name, a, b in zip(("train_loss", "train_acc", "train_f1", "valid_loss", "valid_acc", "valid_f1"), h_hip, h_ref):
        assert len(a) == len(b) == 3
        if "loss" in name:
            assert _close(a, b, 2e-3), (name, a, b)
        else:
            assert _close(a, b, 1e-12), (name, a, b)
    assert np.all(np.isfinite(w_hip)) and np.abs(w_hip - w_ref).max() <= 5e-2, (w_hip, w_ref)
    assert m.use_stream == "multi-GB"
    m.load_state_dict(torch.load(str(tmp_path / "hip_best.pt"), weights_only=True))
    m.load_state_dict(torch.load(str(tmp_path / "hip_last.pt"), weights_only=True))


@pytest.mark.gpu
def test_gb_loops_run_with_the_fused_optimizer_and_loss_on_gpu(golden_dir, tmp_path):
    """The production pairing: ClipAdamW (fused clip + update) and the fused focal-loss kernel through GB_estimate and one
    train_GB_dynamic epoch; invariants only (finite histories, weights sum to 1, parameters moved)."""
    from oracle.fake_multimodal import clip_loaders
    from src.loss import FocalLoss
    from src.optim import ClipAdamW
    tr, va = clip_loaders()
    m, _ = _native_and_oracle(golden_dir)
    before = [p.detach().clone() for p in m.parameters()]
    opt = ClipAdamW(m.parameters(), lr=1e-3)
    mk = lambda: FocalLoss(weight=torch.tensor([1.0, 2.0], device="cuda:0"), gamma=2.0)
    last = str(tmp_path / "last.pt")
    torch.save(m.state_dict(), last)
    w = GB_estimate(2, tr, va, last, m, opt, None, mk(), "cuda:0", 1.0)
    assert abs(sum(w.values()) - 1.0) < 1e-9
    loss_gb = GradientBlending(mk(), mk(), mk(), 0.2, 0.3, 0.5, 1.0)
    hist = train_GB_dynamic(tr, va, m, opt, None, loss_gb, mk(), "cuda:0", num_epoch=1, verbose=None,
                            save_best_dir=str(tmp_path / "b.pt"), save_last_dir=last, exp_dir=str(tmp_path / "exp"),
                            max_norm_grad=1.0, criteria="loss")
    assert all(np.isfinite(h).all() for h in hist)
    moved = sum(int(not torch.equal(a, p.detach())) for a, p in zip(before, m.parameters()))
    assert moved >= len(before) - 2

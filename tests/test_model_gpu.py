"""GPU parity of the whole R(2+1)D classifier (executor plan + head + fused focal loss) against the golden
fixtures generated from the reference, and against the oracle at a second shape.

Tolerance (north_star): 1e-3 of each tensor's scale in fp32 against the reference's fp32 CPU outputs.
The reference's own fp32 rounding noise dominates that budget on the tiny fixtures (its gradients sit
~6e-4 from an fp64 evaluation when the batch is 2-3 samples), so a second test compares the HIP path with
the oracle evaluated in fp64 at 1e-4.  Label bookkeeping (pred) is bit-exact.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src.models.R2Plus1D import R2Plus1DClassifier
    from src.loss import FocalLoss

from oracle import losses as ol, r2plus1d as orc, step as ostep

DEV = "cuda:0"
TOL = 1e-3
TOL64 = 1e-4


def subsample(t, n=48):
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride][:n].cpu().numpy()


def close(a, b, tol, floor=1e-6):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b))) / max(floor, float(np.max(np.abs(b)))) <= tol


def build(ls, T, S, alpha, seed):
    model = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=ls, alpha=alpha)
    params, bufs = orc.synth_state(ls, seed, alpha)
    sd = dict(params); sd.update(bufs)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return model.to(DEV)


TAGS = [f"r2p1d_1111_s{i}" for i in (1, 2, 3, 4)] + [f"r2p1d_1221_s{i}" for i in (11, 12, 13)]
MAX_FLIPS = 8            # LeakyReLU inputs that may land on the other side of zero than in the fp64 oracle ...
FLIP_NEAR_ZERO = 1e-4    # ... each within this fraction of its tensor's rms of zero (i.e. within rounding error of the kink)


def _run_fixture(golden_dir, tag, exact=True):
    """Forward parity against the reference fixture, then the gradient comparison described in tests/kink_util.py.
    Returns (worst error vs the fp32 reference fixture, worst vs the fp64 oracle with its own activation pattern,
    worst vs the fp64 oracle on the HIP path's activation pattern, list of sign flips)."""
    from src import ops
    from tests import kink_util as ku
    ops.set_exact_fp32(exact)
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    ls = [int(v) for v in g["layer_sizes"]]
    B, T, S, alpha, seed = int(g["B"]), int(g["T"]), int(g["S"]), float(g["alpha"]), int(g["seed"])
    model = build(ls, T, S, alpha, seed)
    model.train()
    x = orc.synth_clip(B, T, S, seed)
    y = orc.synth_labels(B, seed)
    w = torch.from_numpy(g["weight"]); gamma = float(g["gamma"])
    loss_fn = FocalLoss(weight=w, gamma=gamma)
    pre_hip, _ = ku.hip_preactivations(model, x.to(DEV))             # (throw-away running statistics: model untouched)
    feat = model.res2plus1d(x.to(DEV))
    logits = model.linear(feat)
    loss = loss_fn(logits, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    # ---- forward quantities: continuous in the inputs, strict on EVERY seed
    assert close(feat.detach().cpu().numpy(), g["trunk"], TOL), tag
    assert close(logits.detach().cpu().numpy(), g["logits"], TOL), tag
    assert close(loss.item(), g["loss"], TOL), tag
    print(tag, "exact" if exact else "split", "logit err %.2e" % float(
        np.abs(logits.detach().cpu().numpy() - g["logits"]).max() / np.abs(g["logits"]).max()))
    ref_pred = torch.softmax(torch.from_numpy(g["logits"]), 1).max(1)[1]
    assert torch.equal(loss_fn.last_pred.cpu(), ref_pred), tag          # bit-exact bookkeeping
    sd = model.state_dict()
    for k in [k for k in g.files if k.startswith("buf/")]:
        assert close(sd[k[4:]].cpu().numpy(), g[k], TOL), (tag, k)
    assert int(sd["linear.1.num_batches_tracked"]) == int(g["nbt"])
    assert int(sd["res2plus1d.conv1.spatio_conv.bn.num_batches_tracked"]) == 1
    # ---- gradients
    named = dict(model.named_parameters())
    gmax = max(float(g["gnorm/" + str(k)]) for k in g["param_names"])
    params, bufs = orc.synth_state(ls, seed, alpha)
    lossf = lambda o, t: ol.focal_loss(o, t, w.double(), gamma)
    _, _, g64, pre64 = ku.oracle_preactivations(x, y, params, bufs, ls, alpha, lossf)
    flips = ku.flips(pre_hip, pre64)
    params, bufs = orc.synth_state(ls, seed, alpha)
    _, _, g64p = ku.oracle_grads_on_pattern(x, y, params, bufs, ls, alpha, lossf, ku.sign_masks(pre_hip)) if flips else (None, None, g64)
    worst32 = worst64 = worst64p = 0.0
    for k in g["param_names"]:
        k = str(k)
        gr = named[k].grad
        assert gr is not None, k
        if k == "linear.0.bias":    # bias in front of BatchNorm1d: analytically zero
            assert float(gr.abs().max()) < 1e-5 * gmax
            continue
        ref = g["gsub/" + k].astype(np.float64)
        sc = max(float(np.abs(ref).max()), 1e-5 * gmax)
        worst32 = max(worst32, float(np.abs(subsample(gr).astype(np.float64) - ref).max()) / sc)
        for which, gg in (("own", g64), ("hip", g64p)):
            r64 = gg[k].numpy()
            sc64 = max(float(np.abs(r64).max()), 1e-5 * gmax)
            e = float(np.abs(gr.cpu().numpy().astype(np.float64) - r64).max()) / sc64
            if which == "own":
                worst64 = max(worst64, e)
            else:
                worst64p = max(worst64p, e)
    return worst32, worst64, worst64p, flips


def _assert_gradients(tag, g, w32, w64, w64p, flips):
    """The HIP gradients are the exact gradients (1e-3) of the function the HIP forward evaluated; that function differs
    from the fp64 oracle's only in which side of zero a few near-zero LeakyReLU inputs fell on."""
    for name, idx, vh, vo, rms in flips:
        print(f"  sign flip {tag}: {name}[{idx}]  hip {vh:+.3e}  fp64 oracle {vo:+.3e}  (tensor rms {rms:.3e})")
    print(tag, "grads: vs ref fp32 fixture %.2e, vs fp64 oracle %.2e, vs fp64 oracle on the HIP activation pattern %.2e, "
          "flips %d, ref_noise %.2e" % (w32, w64, w64p, len(flips), float(g["ref_noise"])))
    assert len(flips) <= MAX_FLIPS, (tag, len(flips))
    for name, idx, vh, vo, rms in flips:
        assert abs(vh) <= FLIP_NEAR_ZERO * rms and abs(vo) <= FLIP_NEAR_ZERO * rms, (tag, name, idx, vh, vo, rms)
    assert w64p < TOL, (tag, w64p)                      # no escape hatch: every fixture, both arithmetic modes
    if not flips:
        assert w64 < TOL, (tag, w64)
        if float(g["ref_noise"]) < 2e-4:                # the reference's own fp32 run is kink-free too: hold it as well
            assert w32 < TOL, (tag, w32)


@pytest.mark.parametrize("tag", TAGS)
def test_classifier_matches_reference(golden_dir, tag):
    """Exact-fp32 arithmetic mode.  Forward parity (features, logits, loss, running statistics: 1e-3 against the reference's
    fixture; predictions bit-exact) on every fixture.  Gradients: 1e-3 against the fp64 oracle evaluated on the activation
    pattern the HIP forward actually took; sign differences with the oracle's own pattern are listed, must be few and must
    sit within rounding error of zero (tests/kink_util.py)."""
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    try:
        w32, w64, w64p, flips = _run_fixture(golden_dir, tag, exact=True)
    finally:
        from src import ops
        ops.set_exact_fp32(False)
    _assert_gradients(tag, g, w32, w64, w64p, flips)


@pytest.mark.parametrize("tag", TAGS)
def test_classifier_split_mode(golden_dir, tag):
    """Default arithmetic mode (md_set_exact_fp32(0): fp16 / bf16 hi+lo split products): same assertions."""
    g = np.load(os.path.join(golden_dir, tag + ".npz"))
    try:
        w32, w64, w64p, flips = _run_fixture(golden_dir, tag, exact=False)
    finally:
        from src import ops
        ops.set_exact_fp32(False)
    _assert_gradients(tag, g, w32, w64, w64p, flips)


def test_state_dict_keys_match_reference_layout():
    model = R2Plus1DClassifier(input_size=(3, 4, 32, 32), num_classes=2, layer_sizes=[1, 2, 2, 1], alpha=0.01)
    keys = set(model.state_dict().keys())
    want = set(orc.param_shapes([1, 2, 2, 1]).keys()) | set(orc.buffer_shapes([1, 2, 2, 1]).keys())
    assert keys == want
    assert sum(p.numel() for p in model.parameters()) == 1587523


def test_eval_mode_and_oracle_second_shape():
    ls, B, T, S, alpha, seed = [1, 1, 1, 1], 3, 4, 36, 0.05, 21
    model = build(ls, T, S, alpha, seed)
    params, bufs = orc.synth_state(ls, seed, alpha)
    x = orc.synth_clip(B, T, S, seed)
    # one training step moves the running statistics on both sides, then compare eval-mode logits
    model.train()
    model(x.to(DEV))
    orc.classifier_forward(x, params, bufs, ls, alpha, training=True)
    model.eval()
    with torch.no_grad():
        out = model(x.to(DEV))
        enc = model.encode(x.to(DEV))
    ref = orc.classifier_forward(x, params, bufs, ls, alpha, training=False)
    ref_enc = orc.trunk_forward(x, params, bufs, ls, alpha, training=False)
    torch.cuda.synchronize()
    assert close(out.cpu().numpy(), ref.numpy(), TOL)
    assert close(enc.cpu().numpy(), ref_enc.numpy(), TOL)


def test_cpu_input_fails_loudly():
    model = R2Plus1DClassifier(input_size=(3, 4, 32, 32), num_classes=2, layer_sizes=[1, 1, 1, 1], alpha=0.01).to(DEV)
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 4, 32, 32))

"""GPU parity at the BASELINE.json shape (T=21, 128x128, layer_sizes [1,2,2,1]) against the oracle run on the host.

Forward: logits / loss within 1e-3 of the fp32 oracle in both arithmetic modes.  Gradients: relative L2 error per
parameter against the oracle evaluated in fp64, bounded by a small multiple of the fp32 oracle's own distance to
fp64 (the yardstick for what fp32 arithmetic can deliver on this workload).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src import ops
    from src.models.R2Plus1D import R2Plus1DClassifier
    from src.loss import FocalLoss

from oracle import losses as ol, r2plus1d as orc, step as ostep

DEV = "cuda:0"
LS, B, T, S, ALPHA, SEED = [1, 2, 2, 1], 4, 21, 128, 0.01, 77


@pytest.fixture(scope="module")
def reference():
    torch.set_num_threads(16)
    params, bufs = orc.synth_state(LS, SEED, ALPHA)
    x = orc.synth_clip(B, T, S, SEED); y = orc.synth_labels(B, SEED)
    w = torch.ones(2)
    logits, loss, grads = ostep.r2plus1d_loss_and_grads(x, y, params, bufs, LS, ALPHA, lambda o, t: ol.focal_loss(o, t, w, 2.0))
    # the same oracle in fp64: the yardstick for how much of a deviation is fp32 rounding noise of the CPU path itself
    p64 = {k: v.double() for k, v in params.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in orc.synth_state(LS, SEED, ALPHA)[1].items()}
    _, _, g64 = ostep.r2plus1d_loss_and_grads(x.double(), y, p64, b64, LS, ALPHA, lambda o, t: ol.focal_loss(o, t, w.double(), 2.0))
    return x, y, logits, loss, grads, g64


@pytest.mark.parametrize("exact", [False, True], ids=["split", "exact_fp32"])
def test_fullsize_forward_backward(reference, exact):
    x, y, ref_logits, ref_loss, ref_g, g64 = reference
    ops.set_exact_fp32(exact)
    try:
        model = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=LS, alpha=ALPHA)
        params, bufs = orc.synth_state(LS, SEED, ALPHA)
        sd = dict(params); sd.update(bufs)
        model.load_state_dict(sd, strict=True)
        model.to(DEV).train()
        loss_fn = FocalLoss(weight=torch.ones(2), gamma=2.0)
        logits = model(x.to(DEV))
        loss = loss_fn(logits, y.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        from tests import kink_util as ku
        rep = ku.gradient_report(model, x, y, LS, ALPHA, SEED, torch.ones(2), 2.0, DEV)      # (same arithmetic mode)
    finally:
        ops.set_exact_fp32(False)
    lerr = float((logits.detach().cpu() - ref_logits).abs().max() / ref_logits.abs().max())
    assert lerr < 1e-3, lerr
    assert abs(loss.item() - float(ref_loss)) < 1e-3 * max(1.0, abs(float(ref_loss)))
    worst, worst_name, errs, errs32, cpu_noise = 0.0, "", [], [], []
    gmax = max(float(v.norm()) for v in g64.values())
    for k, p in model.named_parameters():
        if k == "linear.0.bias":
            continue
        r = g64[k]
        den = max(float(r.norm()), 1e-6 * gmax)
        e = float((p.grad.cpu().double() - r).norm() / den)
        errs.append(e)
        errs32.append(float((p.grad.cpu() - ref_g[k]).norm() / den))
        cpu_noise.append(float((ref_g[k].double() - r).norm() / den))
        if e > worst:
            worst, worst_name = e, k
    print(f"mode={'exact' if exact else 'split'} logits relerr {lerr:.2e}; gradient rel-L2 vs fp64 oracle: median "
          f"{np.median(errs):.2e} worst {worst:.2e} ({worst_name}); vs fp32 oracle: median {np.median(errs32):.2e}; "
          f"fp32 oracle vs fp64 oracle: median {np.median(cpu_noise):.2e} worst {max(cpu_noise):.2e}")

    # Gradients: the yardstick is the fp64 oracle evaluated on the activation pattern the HIP forward took
    # (tests/kink_util.py); no multiple-of-the-CPU-noise allowance.
    for name, idx, vh, vo, rms in rep["flips"][:12]:
        print(f"  sign flip: {name}[{idx}]  hip {vh:+.3e}  fp64 oracle {vo:+.3e}  (tensor rms {rms:.3e})")
    print("B=%d mode=%s: vs fp64 oracle on the HIP activation pattern: worst %.2e (%s), median %.2e; flips %d" % (
        B, "exact" if exact else "split", rep["worst_pattern"], rep["worst_name"], rep["median_pattern"], len(rep["flips"])))
    assert len(rep["flips"]) <= rep["max_flips"], (len(rep["flips"]), rep["elements"])
    for name, idx, vh, vo, rms in rep["flips"]:
        assert abs(vh) <= 1e-4 * rms and abs(vo) <= 1e-4 * rms, (name, idx, vh, vo, rms)
    assert rep["worst_pattern"] < 1e-3, rep


def test_headline_shape_b8_gradients_against_fp64_oracle():
    """The shape bench.py times (B=8, T=21, 128x128, [1,2,2,1], default split arithmetic): logits / loss within 1e-3 of the
    fp64 oracle, every parameter gradient within 1e-3 (median) / 3e-3 (worst, relative L2) of the fp64 oracle on the HIP
    path's activation pattern; flips listed, few, and within rounding error of zero."""
    from tests import kink_util as ku
    torch.set_num_threads(16)
    B8, seed = 8, 1234
    params, bufs = orc.synth_state(LS, seed, ALPHA)
    x = orc.synth_clip(B8, T, S, seed); y = orc.synth_labels(B8, seed)
    model = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=LS, alpha=ALPHA)
    sd = dict(params); sd.update(bufs)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train()
    logits = model(x.to(DEV))
    loss = FocalLoss(weight=torch.ones(2), gamma=2.0)(logits, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    p64 = {k: v.double() for k, v in params.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in orc.synth_state(LS, seed, ALPHA)[1].items()}
    w = torch.ones(2, dtype=torch.float64)
    l64, L64, _ = ostep.r2plus1d_loss_and_grads(x.double(), y, p64, b64, LS, ALPHA, lambda o, t: ol.focal_loss(o, t, w, 2.0))
    assert float((logits.detach().cpu().double() - l64).abs().max() / l64.abs().max()) < 1e-3
    assert abs(loss.item() - float(L64)) < 1e-3 * max(1.0, abs(float(L64)))
    rep = ku.gradient_report(model, x, y, LS, ALPHA, seed, torch.ones(2), 2.0, DEV)
    for name, idx, vh, vo, rms in rep["flips"][:12]:
        print(f"  sign flip: {name}[{idx}]  hip {vh:+.3e}  fp64 oracle {vo:+.3e}  (tensor rms {rms:.3e})")
    print("B=8 headline shape: vs fp64 oracle own pattern %.2e; on the HIP activation pattern: worst %.2e (%s), median %.2e; flips %d" % (
        rep["worst_own"], rep["worst_pattern"], rep["worst_name"], rep["median_pattern"], len(rep["flips"])))
    assert len(rep["flips"]) <= rep["max_flips"], (len(rep["flips"]), rep["elements"])
    for name, idx, vh, vo, rms in rep["flips"]:
        assert abs(vh) <= 1e-4 * rms and abs(vo) <= 1e-4 * rms, (name, idx, vh, vo, rms)
    assert rep["worst_pattern"] < 1e-3, rep


def test_side_stream_schedule_is_bit_identical_to_serial():
    """Side-stream schedule (forward: skip-path convolutions of the downsampling blocks; backward: weight gradients;
    default) vs everything on one stream: same kernels, so every gradient must be bit-identical -- a gradient buffer overwritten before its weight gradient has read it (the hazard
    the events guard against) would show up here.  Three repetitions per schedule, BASELINE shape, B=8."""
    model = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=LS, alpha=ALPHA)
    params, bufs = orc.synth_state(LS, SEED, ALPHA)
    sd = dict(params); sd.update(bufs)
    model.load_state_dict(sd, strict=True)
    model.to(DEV).train()
    loss_fn = FocalLoss(weight=torch.ones(2), gamma=2.0)
    x = orc.synth_clip(8, T, S, SEED + 1).to(DEV); y = orc.synth_labels(8, SEED + 1).to(DEV)

    with torch.no_grad():
        model(x)                                  # creates the plan for this shape

    def grads(side):
        out = []
        for _ in range(3):
            model.zero_grad(set_to_none=True)
            for plan in model.res2plus1d._plans.values():
                plan.use_side_stream(side)        # forward (skip-path convolutions) and backward (weight gradients)
            logits = model(x)
            loss_fn(logits, y).backward()
            torch.cuda.synchronize()
            out.append([p.grad.detach().clone() for p in model.parameters()])
        return out

    serial = grads(False)
    side = grads(True)
    for run in serial[1:] + side:
        for a, b in zip(serial[0], run):
            assert torch.equal(a, b)

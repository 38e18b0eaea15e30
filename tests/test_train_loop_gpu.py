"""GPU: the mirrored epoch loops (src.train) against the fixture recorded from the reference's train_per_epoch
(tests/golden/step_tiny.npz: 3 fixed batches, AdamW 2e-4, clip 1.0, FocalLoss) -- loss within 1e-3, predictions,
accuracy and macro-F1 bit-exact, parameter deltas within 2e-3 of the AdamW step size -- plus valid_per_epoch,
GradientBlending and train_DRW plumbing."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src import ops
    from src.GradientBlending import GradientBlending
    from src.loss import CELoss, FocalLoss, LDAMLoss
    from src.models.R2Plus1D import R2Plus1DClassifier
    from src.train import train_DRW, train_per_epoch, valid_per_epoch

from oracle import losses as ol, r2plus1d as orc

DEV = "cuda:0"


def subsample(t, n=48):
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride][:n].cpu().numpy()


def _model(ls, T, S, alpha, seed):
    m = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=ls, alpha=alpha)
    params, bufs = orc.synth_state(ls, seed, alpha)
    sd = dict(params); sd.update(bufs)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV)


@pytest.mark.parametrize("fused_opt", [False, True], ids=["torch_adamw", "clip_adamw"])
@pytest.mark.parametrize("exact", [True, False], ids=["exact_fp32", "split"])
def test_train_per_epoch_matches_reference_step_fixture(golden_dir, exact, fused_opt):
    g = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    ls = [int(v) for v in g["layer_sizes"]]
    B, T, S, alpha, seed = int(g["B"]), int(g["T"]), int(g["S"]), float(g["alpha"]), int(g["seed"])
    ops.set_exact_fp32(exact)
    try:
        model = _model(ls, T, S, alpha, seed)
        before = {k: v.detach().clone() for k, v in model.named_parameters()}
        batches = [(orc.synth_clip(B, T, S, seed + i), orc.synth_labels(B, seed + i, 0.4)) for i in range(3)]
        if fused_opt:      # src.optim.ClipAdamW: clip_grad_norm_ + AdamW through md_opt_* (same fixture, same bars)
            from src.optim import ClipAdamW
            opt = ClipAdamW(model.parameters(), lr=2e-4)
        else:
            opt = torch.optim.AdamW(model.parameters(), lr=2e-4)
        loss_fn = FocalLoss(weight=torch.tensor([1.0, 1.0]), gamma=2.0)
        preds = []
        hook = model.register_forward_hook(lambda m, i, o: preds.append(torch.softmax(o, 1).max(1)[1].cpu().numpy().copy()))
        tl, ta, tf = train_per_epoch(batches, model, opt, None, loss_fn, DEV, 1.0, "single")
        hook.remove()
    finally:
        ops.set_exact_fp32(False)
    assert np.array_equal(np.stack(preds), g["preds"])                 # bit-exact label bookkeeping
    assert ta == float(g["train_acc"])
    assert abs(tf - float(g["train_f1"])) < 1e-12
    # epoch loss over the 3-step trajectory (mean of the three step losses)
    dloss = abs(tl - float(g["train_loss"])) / max(1.0, abs(float(g["train_loss"])))
    print("epoch-loss deviation from the reference fixture: %.2e (%s)" % (dloss, "exact" if exact else "split"))
    # Parameter deltas.  AdamW's first steps move every element by ~lr * sign(gradient): where a gradient element is at
    # round-off level its sign -- and with it a full 2 lr -- depends on the last bits of ANY implementation (the
    # reference's fp32 run included).  So the comparison is made where the gradient is resolved: elements whose first-step
    # gradient (fp64 oracle) exceeds 1e-3 of their tensor's largest.  Those must agree with the reference's recorded
    # deltas to 0.05 lr in both modes; the excluded share is reported and bounded.
    from oracle import step as ostep
    params, bufs = orc.synth_state(ls, seed, alpha)
    p64 = {k: v.double() for k, v in params.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in bufs.items()}
    w64 = torch.ones(2, dtype=torch.float64)
    _, _, g64 = ostep.r2plus1d_loss_and_grads(batches[0][0].double(), batches[0][1], p64, b64, ls, alpha,
                                              lambda o, t: ol.focal_loss(o, t, w64, 2.0))
    errs, resolved = [], []
    for k, p in model.named_parameters():
        if k == "linear.0.bias":        # feeds BatchNorm1d: its gradient is analytically zero, Adam's step of it pure noise
            continue
        gs = np.abs(subsample(g64[k]))
        resolved.append(gs > 1e-3 * float(g64[k].abs().max()))
        errs.append(np.abs(subsample(p.detach() - before[k]) - g["dsub/" + k]) / 2e-4)
    errs = np.concatenate(errs); resolved = np.concatenate(resolved)
    print("parameter-delta error / lr on resolved-gradient elements: worst %.3f median %.4f; unresolved share %.3f (their worst %.3f)" % (
        errs[resolved].max(), float(np.median(errs[resolved])), 1.0 - float(resolved.mean()),
        errs[~resolved].max() if (~resolved).any() else 0.0))
    assert float(resolved.mean()) > 0.85
    if exact:
        # exact-fp32 arithmetic reproduces the reference's recorded 3-step trajectory itself
        assert errs[resolved].max() < 0.05, float(errs[resolved].max())
        assert dloss < 1e-3, dloss
    else:
        # The default split arithmetic lands one LeakyReLU input of this fixture on the other side of zero than the
        # reference did (listed by test_split_mode_single_step_matches_oracle_on_its_activation_pattern below), which moves
        # the first AdamW step of many elements by O(lr) and the following losses with it; the trajectory-level bars are the
        # loose ones here, the step itself is held to the strict bar in that test.
        assert dloss < 2e-3, dloss
        assert float(np.median(errs[resolved])) < 0.05


def test_split_mode_single_step_matches_oracle_on_its_activation_pattern(golden_dir):
    """One optimisation step of train_per_epoch (default split arithmetic, clip 1.0, ClipAdamW 2e-4) on the first batch of the
    step fixture: every parameter's delta against the fp64 oracle's step -- gradients evaluated on the activation pattern the
    HIP forward took (tests/kink_util.py), then clip_grad_norm + AdamW in fp64 -- within 0.05 lr wherever the gradient is
    resolved (> 1e-3 of its tensor's largest).  The sign flips against the oracle's own pattern are listed and must be
    near-zero pre-activations."""
    from oracle import step as ostep
    from src.optim import ClipAdamW
    from tests import kink_util as ku
    g = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    ls = [int(v) for v in g["layer_sizes"]]
    B, T, S, alpha, seed = int(g["B"]), int(g["T"]), int(g["S"]), float(g["alpha"]), int(g["seed"])
    lr = 2e-4
    model = _model(ls, T, S, alpha, seed)
    x, y = orc.synth_clip(B, T, S, seed), orc.synth_labels(B, seed, 0.4)
    model.train()
    pre_hip, _ = ku.hip_preactivations(model, x.to(DEV))
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt = ClipAdamW(model.parameters(), lr=lr)
    train_per_epoch([(x, y)], model, opt, None, FocalLoss(weight=torch.ones(2), gamma=2.0), DEV, 1.0, "single")
    torch.cuda.synchronize()
    w64 = torch.ones(2, dtype=torch.float64)
    lossf = lambda o, t: ol.focal_loss(o, t, w64, 2.0)
    params, bufs = orc.synth_state(ls, seed, alpha)
    _, _, _, pre64 = ku.oracle_preactivations(x, y, params, bufs, ls, alpha, lossf)
    fl = ku.flips(pre_hip, pre64)
    for name, idx, vh, vo, rms in fl:
        print(f"  sign flip: {name}[{idx}]  hip {vh:+.3e}  fp64 oracle {vo:+.3e}  (tensor rms {rms:.3e})")
        assert abs(vh) <= 1e-4 * rms and abs(vo) <= 1e-4 * rms
    assert len(fl) <= 8
    params, bufs = orc.synth_state(ls, seed, alpha)
    _, _, g64 = ku.oracle_grads_on_pattern(x, y, params, bufs, ls, alpha, lossf, ku.sign_masks(pre_hip))
    # the reference's step (src/train.py:64-66) in fp64: clip_grad_norm_(1.0), AdamW(lr, betas (0.9, 0.999), eps 1e-8, wd 1e-2)
    names = [k for k, _ in model.named_parameters()]
    leaves = [params[k].double().clone().requires_grad_(True) for k in names]
    for leaf, k in zip(leaves, names):
        leaf.grad = g64[k].clone()
    ostep.clip_grad_norm([l.grad for l in leaves], 1.0)
    torch.optim.AdamW(leaves, lr=lr).step()
    worst, excluded, total = 0.0, 0, 0
    for leaf, k in zip(leaves, names):
        if k == "linear.0.bias":
            continue
        d_ref = (leaf.detach() - params[k].double())
        d_hip = (dict(model.named_parameters())[k].detach() - before[k]).cpu().double()
        res = g64[k].abs() > 1e-3 * g64[k].abs().max()
        total += res.numel(); excluded += int((~res).sum())
        if res.any():
            worst = max(worst, float(((d_hip - d_ref).abs()[res]).max()) / lr)
    print("single step, split mode: worst delta error on resolved-gradient elements %.4f lr; unresolved share %.3f; flips %d" % (
        worst, excluded / total, len(fl)))
    assert excluded / total < 0.25
    assert worst < 0.05, worst



def test_valid_per_epoch_and_other_losses():
    ls, B, T, S, alpha, seed = [1, 1, 1, 1], 4, 4, 32, 0.01, 9
    model = _model(ls, T, S, alpha, seed)
    params, bufs = orc.synth_state(ls, seed, alpha)
    batches = [(orc.synth_clip(B, T, S, seed + i), orc.synth_labels(B, seed + i, 0.4)) for i in range(2)]
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    w = torch.tensor([1.7, 0.3])
    m = ol.ldam_margins([100, 2000], 0.5)
    for loss_fn, ref_fn in (
            (CELoss(weight=w), lambda o, t: ol.ce_loss(o, t, w)),
            (LDAMLoss([100, 2000], max_m=0.5, weight=w, s=30), lambda o, t: ol.ldam_loss(o, t, m, w, 30.0)),
            (FocalLoss(weight=w, gamma=0.5), lambda o, t: ol.focal_loss(o, t, w, 0.5))):
        vl, va, vf = valid_per_epoch(batches, model, opt, loss_fn, DEV, "single")
        tot, correct, n = 0.0, 0, 0
        for x, y in batches:
            out = orc.classifier_forward(x, params, {k: v.clone() for k, v in bufs.items()}, ls, alpha, training=False)
            tot += float(ref_fn(out, y)); correct += int((out.argmax(1) == y).sum()); n += B
        assert abs(vl - tot / n) < 1e-3 * max(1.0, abs(tot / n)), type(loss_fn).__name__
        assert va == correct / n


def test_gradient_blending_against_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    y = torch.from_numpy(g["gb/y"]).to(DEV)
    xs = {n: torch.from_numpy(g[f"gb/x_{n}"]).to(DEV).requires_grad_(True) for n in ("multi", "vis", "ts")}
    one = torch.ones(2)
    gb = GradientBlending(FocalLoss(one, 2.0), FocalLoss(one, 2.0), FocalLoss(one, 2.0), 0.1, 0.4, 0.5)
    L = gb(xs["multi"], xs["vis"], xs["ts"], y)
    L.backward()
    assert abs(L.item() - float(g["gb/L"])) < 1e-5
    for n in xs:
        assert float((xs[n].grad.cpu() - torch.from_numpy(g[f"gb/g_{n}"])).abs().max()) < 1e-5
    gb.update_weights({"video": 1.0, "0D": 0.0, "multi": 0.0})
    assert abs(gb(xs["multi"], xs["vis"], xs["ts"], y).item() - float(ol.focal_loss(xs["vis"].detach().cpu(), y.cpu(), one))) < 1e-5


def test_train_drw_updates_class_weights(tmp_path):
    ls, B, T, S, alpha, seed = [1, 1, 1, 1], 4, 4, 32, 0.01, 3
    model = _model(ls, T, S, alpha, seed)
    batches = [(orc.synth_clip(B, T, S, seed), orc.synth_labels(B, seed, 0.4))]
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4)
    seen = []

    class Rec(FocalLoss):
        def update_weight(self, weight=None):
            seen.append(weight.detach().cpu().numpy().copy()); super().update_weight(weight)

    out = train_DRW(batches, batches, model, opt, Rec(torch.ones(2), 2.0), DEV, num_epoch=4, verbose=0,
                    save_best_dir=str(tmp_path / "b.pt"), save_last_dir=str(tmp_path / "l.pt"), exp_dir=None,
                    max_norm_grad=1.0, cls_num_list=[100, 2000], betas=[0, 0.25, 0.75, 0.9])
    assert len(out) == 6 and len(out[0]) == 4 and len(seen) == 4
    ref = np.stack([ol.drw_weights(e, 4, [0, 0.25, 0.75, 0.9], [100, 2000]) for e in range(4)])
    assert np.array_equal(np.stack(seen), ref)
    sd = torch.load(str(tmp_path / "l.pt"), weights_only=True)
    assert set(sd) == set(model.state_dict())

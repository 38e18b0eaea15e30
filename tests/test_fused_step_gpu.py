"""GPU: the one-call training step (md_plan_train_step, src/_step.py::FusedTrainStep) against the composed step it replaces
(reference loop src/train.py:40-66; module forward src/models/R2Plus1D.py:262-265) -- same kernels in the same order, so every
parameter, buffer, gradient and optimizer moment must be BIT-identical -- and, through train_per_epoch, against the reference's
recorded step fixture (tests/golden/step_tiny.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _imports():
    from src.models.R2Plus1D import R2Plus1DClassifier
    from src.loss import CELoss, FocalLoss, LDAMLoss
    from src.optim import ClipAdamW
    from src._step import FusedTrainStep, applicable
    return R2Plus1DClassifier, FocalLoss, LDAMLoss, CELoss, ClipAdamW, FusedTrainStep, applicable


def _pair(ls, T, S, seed):
    R2Plus1DClassifier = _imports()[0]
    torch.manual_seed(seed)
    a = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=ls, alpha=0.01).to(DEV).train()
    b = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=ls, alpha=0.01).to(DEV).train()
    b.load_state_dict(a.state_dict())
    return a, b


def _composed_step(model, loss_fn, opt, x, y, max_norm):
    opt.zero_grad()
    out = model(x)
    loss = loss_fn(out, y)
    loss.backward()
    opt.step(max_norm=max_norm)
    return loss.detach(), out.detach(), loss_fn.last_pred


@pytest.mark.parametrize("loss_kind", ["focal", "ldam", "ce"])
def test_fused_step_is_bit_identical_to_the_composed_step(loss_kind):
    R2Plus1DClassifier, FocalLoss, LDAMLoss, CELoss, ClipAdamW, FusedTrainStep, applicable = _imports()
    ls, B, T, S = [1, 2, 1, 1], 4, 6, 32
    ma, mb = _pair(ls, T, S, 11)

    def mk():
        w = torch.tensor([1.0, 2.5])
        if loss_kind == "focal":
            return FocalLoss(weight=w, gamma=2.0)
        if loss_kind == "ldam":
            return LDAMLoss([30, 10], max_m=0.5, weight=w, s=30)
        return CELoss(weight=w)
    la, lb = mk(), mk()
    oa, ob = ClipAdamW(ma.parameters(), lr=2e-4), ClipAdamW(mb.parameters(), lr=2e-4)
    assert applicable(mb, lb, ob)
    fs = FusedTrainStep(mb, lb, ob)
    g = torch.Generator().manual_seed(5)
    for i in range(3):
        x = torch.randn(B, 3, T, S, S, generator=g).to(DEV)
        y = torch.randint(0, 2, (B,), generator=g).to(DEV)
        l0, o0, p0 = _composed_step(ma, la, oa, x, y, 1.0)
        l1, o1, p1, ok = fs(x, y, max_norm=1.0)
        assert float(ok) == 1.0
        assert torch.equal(l0.view(()), l1) and torch.equal(o0, o1) and torch.equal(p0.view(-1), p1.view(-1)), i
        assert torch.equal(lb.last_pred.view(-1), p1.view(-1))
        for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
            assert torch.equal(pa.grad, pb.grad), (i, "grad", k)
            assert torch.equal(pa.detach(), pb.detach()), (i, "param", k)
            assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"]), (i, "exp_avg", k)
            assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"]), (i, "exp_avg_sq", k)
            assert oa.state[pa]["step"] == ob.state[pb]["step"] == i + 1
        for (k, ba), (_, bb) in zip(ma.named_buffers(), mb.named_buffers()):
            assert torch.equal(ba, bb), (i, "buffer", k)
        assert torch.equal(oa.last_grad_norm, ob.last_grad_norm)
    # the state dicts are interchangeable afterwards
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"] == sb["param_groups"]


def test_fused_step_skips_the_update_on_a_non_finite_loss_and_takes_the_step_count_back():
    R2Plus1DClassifier, FocalLoss, LDAMLoss, CELoss, ClipAdamW, FusedTrainStep, applicable = _imports()
    ls, B, T, S = [1, 1, 1, 1], 2, 4, 32
    m, _ = _pair(ls, T, S, 3)
    lf = FocalLoss(weight=torch.tensor([1.0, 1.0]), gamma=2.0)
    opt = ClipAdamW(m.parameters(), lr=1e-3)
    fs = FusedTrainStep(m, lf, opt)
    x = torch.randn(B, 3, T, S, S, device=DEV); y = torch.tensor([0, 1], device=DEV)
    _, _, _, ok = fs(x, y, max_norm=1.0)
    assert float(ok) == 1.0
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    mom = {k: opt.state[p]["exp_avg"].clone() for k, p in m.named_parameters()}
    lf.update_weight(torch.tensor([float("nan"), float("nan")]))
    loss, _, _, ok = fs(x, y, max_norm=1.0)
    assert float(ok) == 0.0 and not bool(torch.isfinite(loss))
    for k, p in m.named_parameters():
        assert torch.equal(p.detach(), before[k]), k
        assert torch.equal(opt.state[p]["exp_avg"], mom[k]), k
    torch.cuda.synchronize()
    sd = opt.state_dict()                         # settles the pending flags
    assert all(int(s["step"]) == 1 for s in sd["state"].values())
    lf.update_weight(torch.tensor([1.0, 1.0]))
    _, _, _, ok = fs(x, y, max_norm=1.0)
    assert float(ok) == 1.0
    assert all(int(opt.state[p]["step"]) == 2 for p in m.parameters())


def test_hooks_and_foreign_optimizers_fall_back_to_the_composed_step():
    R2Plus1DClassifier, FocalLoss, LDAMLoss, CELoss, ClipAdamW, FusedTrainStep, applicable = _imports()
    m, _ = _pair([1, 1, 1, 1], 4, 32, 3)
    lf = FocalLoss(weight=torch.ones(2), gamma=2.0)
    assert applicable(m, lf, ClipAdamW(m.parameters(), lr=1e-3))
    assert not applicable(m, lf, torch.optim.AdamW(m.parameters(), lr=1e-3))
    assert not applicable(m, torch.nn.CrossEntropyLoss(), ClipAdamW(m.parameters(), lr=1e-3))
    assert not applicable(m, lf, ClipAdamW(list(m.parameters())[:-2], lr=1e-3))
    h = m.register_forward_hook(lambda *_: None)
    assert not applicable(m, lf, ClipAdamW(m.parameters(), lr=1e-3))
    h.remove()
    m.eval()
    assert not applicable(m, lf, ClipAdamW(m.parameters(), lr=1e-3))
    with pytest.raises(RuntimeError):
        FusedTrainStep(m, lf, torch.optim.AdamW(m.parameters(), lr=1e-3))


@pytest.mark.parametrize("exact", [True, False], ids=["exact_fp32", "split"])
def test_train_per_epoch_through_the_fused_step_matches_the_reference_fixture(golden_dir, exact, capsys):
    """The loop of tests/test_train_loop_gpu.py without its forward hook (a hook keeps the composed step): predictions, accuracy and
    F1 of the reference's recorded three-step epoch, bit for bit; the epoch loss to the bars of that test; and the same parameters as
    the composed loop (MD_FUSED_STEP=0 semantics, reproduced here by a hook)."""
    from oracle import r2plus1d as orc
    from src import ops, train as T
    R2Plus1DClassifier, FocalLoss, LDAMLoss, CELoss, ClipAdamW, FusedTrainStep, applicable = _imports()
    g = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    ls = [int(v) for v in g["layer_sizes"]]
    B, Tn, S, alpha, seed = int(g["B"]), int(g["T"]), int(g["S"]), float(g["alpha"]), int(g["seed"])

    def mk():
        m = R2Plus1DClassifier(input_size=(3, Tn, S, S), num_classes=2, layer_sizes=ls, alpha=alpha)
        params, bufs = orc.synth_state(ls, seed, alpha)
        sd = dict(params); sd.update(bufs)
        m.load_state_dict(sd, strict=True)
        return m.to(DEV)
    batches = [(orc.synth_clip(B, Tn, S, seed + i), orc.synth_labels(B, seed + i, 0.4)) for i in range(3)]
    ops.set_exact_fp32(exact)
    try:
        mf, mc = mk(), mk()
        of, oc = ClipAdamW(mf.parameters(), lr=2e-4), ClipAdamW(mc.parameters(), lr=2e-4)
        lf, lc = FocalLoss(weight=torch.tensor([1.0, 1.0]), gamma=2.0), FocalLoss(weight=torch.tensor([1.0, 1.0]), gamma=2.0)
        assert T._FUSED_STEPS and T._fused_step(mf, lf, of) is not None
        tl, ta, tf = T.train_per_epoch(batches, mf, of, None, lf, DEV, 1.0, "single")
        hook = mc.register_forward_hook(lambda *_: None)
        assert T._fused_step(mc, lc, oc) is None
        cl, ca, cf = T.train_per_epoch(batches, mc, oc, None, lc, DEV, 1.0, "single")
        hook.remove()
    finally:
        ops.set_exact_fp32(False)
    assert ta == float(g["train_acc"]) and abs(tf - float(g["train_f1"])) < 1e-12
    assert (ta, tf) == (ca, cf)
    dloss = abs(tl - float(g["train_loss"])) / max(1.0, abs(float(g["train_loss"])))
    assert dloss < (1e-3 if exact else 2e-3), dloss
    if exact:
        # the exact-fp32 weight gradient of this debugging mode accumulates with float atomics (csrc/conv_gemm.hip::k_conv_wgrad): two
        # runs of the SAME loop differ in the last bits, so the two loops are compared at that level, not bit for bit
        assert abs(tl - cl) < 1e-5 * max(1.0, abs(cl)), (tl, cl)
        return
    assert tl == cl
    for (k, pa), (_, pb) in zip(mf.named_parameters(), mc.named_parameters()):
        assert torch.equal(pa.detach(), pb.detach()), k
    for (k, ba), (_, bb) in zip(mf.named_buffers(), mc.named_buffers()):
        assert torch.equal(ba, bb), k

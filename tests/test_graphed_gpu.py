"""Whole-step HIP graph helper (src/utils/graphed.py), in-process and AFTER eager default-stream work:
  * tools/graphed_check.py's cases (SlowFast tiny, ViViT with and without dropout, MLSTM_FCN with its CPU-generator noise):
    graph replays bit-identical to eager steps;
  * an eager step on the legacy default stream whose loss tensor has been dropped does not disturb a later capture;
  * with that loss tensor still alive, GraphedStep refuses with a RuntimeError that names the cause (PyTorch's AccumulateGrad
    stream mismatch) instead of letting ROCm segfault in hipStreamEndCapture."""
import os
import runpy
import sys

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_graphed_step_matches_eager_in_process(capsys):
    runpy.run_path(os.path.join(ROOT, "tools", "graphed_check.py"), run_name="__main__")
    assert "graphed_check OK" in capsys.readouterr().out


def _tiny():
    from src.loss import LDAMLoss
    from src.models.slowfast import SlowFast
    torch.manual_seed(0)
    m = SlowFast(input_shape=(3, 8, 64, 64), layers=[1, 1, 1, 1], alpha=4, tau_fast=1, num_classes=2).cuda().train()
    loss_fn = LDAMLoss(cls_num_list=[100, 2000], max_m=0.5, s=1.0, weight=None)
    x = torch.randn(2, 3, 8, 64, 64, device="cuda"); y = torch.tensor([0, 1], device="cuda")
    return m, loss_fn, x, y


def test_capture_after_default_stream_steps():
    from src.utils.graphed import GraphedStep
    m, loss_fn, x, y = _tiny()
    for _ in range(2):                                   # eager, legacy default stream
        m.zero_grad(set_to_none=True)
        loss = loss_fn(m(x), y); loss.backward()
    torch.cuda.synchronize()
    eager_loss = float(loss.detach()); eager = {k: p.grad.clone() for k, p in m.named_parameters()}
    del loss                                             # nothing of the eager graph survives
    gs = GraphedStep(m, loss_fn, [x], y)
    _, l2 = gs([x], y)
    torch.cuda.synchronize()
    assert float(l2.detach()) == eager_loss
    assert all(torch.equal(eager[k], p.grad) for k, p in m.named_parameters())


def test_stale_autograd_graph_is_refused_not_crashed():
    from src.utils.graphed import GraphedStep
    m, loss_fn, x, y = _tiny()
    m.zero_grad(set_to_none=True)
    loss = loss_fn(m(x), y); loss.backward()             # eager, default stream; `loss` stays alive on purpose
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="autograd graph from an earlier training step is still alive"):
        GraphedStep(m, loss_fn, [x], y)
    del loss
    gs = GraphedStep(m, loss_fn, [x], y)                 # and after dropping it the same model captures fine
    gs([x], y); torch.cuda.synchronize()


@pytest.mark.gpu
def test_train_per_epoch_with_graph_replay_matches_the_eager_loop():
    """MD_GRAPH_STEP=1 (src/train.py): every full-size batch of train_per_epoch is one graph replay, the odd-sized last batch and
    the optimizer stay eager.  Two epochs over the same data from the same initial state (SlowFast + MLSTM_FCN, no dropout,
    GradientBlending over LDAM, ClipAdamW): the same kernels on the same bytes, so losses, accuracies, parameters and BatchNorm
    statistics agree bit for bit with the eager loop."""
    from torch.utils.data import DataLoader, Dataset
    import src.train as tr
    from src.GradientBlending import GradientBlending
    from src.loss import LDAMLoss
    from src.models.fusion import FusionGB
    from src.models.MLSTM_FCN import MLSTM_FCN
    from src.models.slowfast import SlowFast
    from src.optim import ClipAdamW

    class Pairs(Dataset):
        def __init__(self):
            g = torch.Generator().manual_seed(5)
            self.v = torch.randn(10, 3, 8, 32, 32, generator=g); self.t = torch.randn(10, 8, 6, generator=g)
            self.y = torch.tensor([0, 1, 0, 1, 1, 0, 0, 1, 1, 0])

        def __len__(self):
            return 10

        def __getitem__(self, i):
            return {"video": self.v[i], "0D": self.t[i]}, self.y[i]

    def make():
        torch.manual_seed(6)
        vis = SlowFast(input_shape=(3, 8, 32, 32), layers=[1, 1, 1, 1], alpha=4, tau_fast=1, num_classes=2, alpha_elu=1.0)
        ts = MLSTM_FCN(n_features=6, fcn_dim=8, kernel_size=3, stride=1, seq_len=8, lstm_dim=8, lstm_n_layers=1, lstm_bidirectional=True,
                       lstm_dropout=0.0, reduction=4, alpha=0.01, n_classes=2)
        ts.noise.std = 0.0      # (the NoiseLayer draws from the CPU generator: the capture's warm-up steps would shift its sequence)
        return FusionGB(2, vis, ts).cuda()

    def run(graph):
        old = tr._GRAPH_STEPS
        tr._GRAPH_STEPS = graph
        try:
            m = make()
            ld = LDAMLoss([100, 2000], max_m=0.5, weight=torch.tensor([1.0, 1.0]).cuda(), s=1.0)
            gb = GradientBlending(ld, ld, ld, 0.1, 0.4, 0.5)
            opt = ClipAdamW(m.parameters(), lr=1e-3, max_norm=1.0)
            loader = DataLoader(Pairs(), batch_size=4, shuffle=False)          # batches of 4, 4, 2
            hist = [tr.train_per_epoch(loader, m, opt, None, gb, "cuda:0", 1.0, "multi-GB") for _ in range(2)]
            used = m.__dict__.get("_md_graphed")
            m.__dict__.pop("_md_graphed", None)
            return hist, {k: v.detach().clone() for k, v in m.state_dict().items()}, used
        finally:
            tr._GRAPH_STEPS = old

    h0, sd0, used0 = run(False)
    h1, sd1, used1 = run(True)
    assert used0 is None and used1 is not None and used1[1] is not None          # the graph was really captured and used
    assert h0 == h1
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


@pytest.mark.gpu
def test_eager_step_of_another_batch_size_leaves_the_captured_noise_buffer_alone():
    """The NoiseLayer of a captured 0D encoder uploads from one pinned staging buffer that GraphedStep refills before each replay.
    An eager step with a different batch size in between (the short last batch of an epoch in the MD_GRAPH_STEP loop) must not
    replace that buffer: afterwards a replay still equals the eager step that draws the same CPU-generator noise."""
    from src.loss import FocalLoss
    from src.models.MLSTM_FCN import MLSTM_FCN
    from src.utils.graphed import GraphedStep
    torch.manual_seed(11)
    m = MLSTM_FCN(n_features=6, fcn_dim=8, kernel_size=3, stride=1, seq_len=8, lstm_dim=8, lstm_n_layers=1, lstm_bidirectional=True,
                  lstm_dropout=0.0, reduction=4, alpha=0.01, n_classes=2).cuda().train()
    m.noise.std = 0.5                                            # make the noise matter
    loss_fn = FocalLoss(torch.tensor([1.0, 1.0]).cuda(), 2.0)
    x = torch.randn(4, 8, 6, device="cuda"); y = torch.tensor([0, 1, 1, 0], device="cuda")
    gs = GraphedStep(m, loss_fn, [x], y)
    static = m.noise.__dict__["_static"]
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                # an eager step with two samples
        m.zero_grad(set_to_none=True)
        loss_fn(m(x[:2]), y[:2]).backward()
    side.synchronize()
    assert m.noise.__dict__["_static"] is static
    gs.bind()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    torch.manual_seed(12)
    _, loss_g = gs([x], y)
    loss_g = float(loss_g); grads_g = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.load_state_dict(sd)
    torch.manual_seed(12)
    with torch.cuda.stream(side):
        m.zero_grad(set_to_none=True)
        loss_e = loss_fn(m(x), y); loss_e.backward()
    side.synchronize()
    assert float(loss_e.detach()) == loss_g
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, grads_g[k]), k
    del loss_e


@pytest.mark.gpu
def test_validation_loop_and_window_curve_with_graph_replay_match_eager():
    """MD_GRAPH_STEP=1 in the inference paths: valid_per_epoch replays the forward of every full-size batch (GraphedForward) and
    video_window_probabilities every full batch of windows -- same kernels on the same bytes, so losses / accuracies / F1 and the
    probability curve agree bit for bit with the eager paths (short last batches run eagerly in both)."""
    from torch.utils.data import DataLoader, Dataset
    import src.train as tr
    import src.utils.prob_curve as pc
    from src.loss import FocalLoss
    from src.models.slowfast import SlowFast
    from src.optim import ClipAdamW

    class Clips(Dataset):
        def __init__(self):
            g = torch.Generator().manual_seed(7)
            self.v = torch.randn(10, 3, 8, 32, 32, generator=g); self.y = torch.arange(10) % 2

        def __len__(self):
            return 10

        def __getitem__(self, i):
            return self.v[i], self.y[i]

    torch.manual_seed(8)
    m = SlowFast(input_shape=(3, 8, 32, 32), layers=[1, 1, 1, 1], alpha=4, tau_fast=1, num_classes=2).cuda()
    loss_fn = FocalLoss(torch.tensor([1.0, 1.0]).cuda(), 2.0)
    opt = ClipAdamW(m.parameters(), lr=1e-3, max_norm=1.0)
    loader = DataLoader(Clips(), batch_size=4, shuffle=False)                   # 4, 4, 2
    frames = torch.randint(0, 256, (40, 40, 40, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(9)).cuda()
    res = {}
    old_t, old_p = tr._GRAPH_STEPS, pc._GRAPH
    try:
        for flag in (False, True):
            tr._GRAPH_STEPS = pc._GRAPH = flag
            val = tr.valid_per_epoch(loader, m, opt, loss_fn, "cuda:0", "single")
            curve = pc.video_window_probabilities(m, frames, 8, 3, crop_size=32, windows_per_launch=4)
            res[flag] = (val, curve)
        assert m.__dict__.get("_md_graphed_eval") not in (None, False) and m.__dict__.get("_md_graphed_curve") not in (None, False)
    finally:
        tr._GRAPH_STEPS, pc._GRAPH = old_t, old_p
        m.__dict__.pop("_md_graphed_eval", None); m.__dict__.pop("_md_graphed_curve", None)
    assert res[False][0] == res[True][0]
    assert np.array_equal(res[False][1][0], res[True][1][0]) and np.array_equal(res[False][1][1], res[True][1][1])

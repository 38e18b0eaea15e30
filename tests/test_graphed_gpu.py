"""Whole-step HIP graph helper (src/utils/graphed.py), in-process and AFTER eager default-stream work:
  * tools/graphed_check.py's cases (SlowFast tiny, ViViT with and without dropout, MLSTM_FCN with its CPU-generator noise):
    graph replays bit-identical to eager steps;
  * an eager step on the legacy default stream whose loss tensor has been dropped does not disturb a later capture;
  * with that loss tensor still alive, GraphedStep refuses with a RuntimeError that names the cause (PyTorch's AccumulateGrad
    stream mismatch) instead of letting ROCm segfault in hipStreamEndCapture."""
import os
import runpy
import sys

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

"""Oracle: the optimisation step and its bookkeeping (test infrastructure).

Reference anchors:
  * train_per_epoch  src/train.py:17-93  zero_grad -> forward -> loss -> finite check -> backward ->
                                         clip_grad_norm_ -> optimizer.step -> argmax bookkeeping -> macro-F1
  * macro F1         sklearn.metrics.f1_score(average="macro") as called at src/train.py:86
  * DP step (derived; the reference's src/distributed.py:29-111 never exchanges gradients):
                     mean over ranks of per-shard gradients, one optimiser step.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence

import numpy as np
import torch

from . import losses, r2plus1d


def predictions(logits: torch.Tensor) -> torch.Tensor:
    """pred = softmax(output).max(1)[1]  (src/train.py:70); int64 (B,)."""
    return torch.softmax(logits, dim=1).max(1)[1]


def macro_f1(labels: np.ndarray, preds: np.ndarray) -> float:
    """Macro F1 over the labels present in y_true or y_pred (sklearn default label set)."""
    classes = np.union1d(labels, preds)
    f = []
    for c in classes:
        tp = float(np.sum((preds == c) & (labels == c)))
        fp = float(np.sum((preds == c) & (labels != c)))
        fn = float(np.sum((preds != c) & (labels == c)))
        d = 2 * tp + fp + fn
        f.append(0.0 if d == 0 else 2 * tp / d)
    return float(np.mean(f))


def clip_grad_norm(grads: Sequence[torch.Tensor], max_norm: float) -> float:
    """torch.nn.utils.clip_grad_norm_ (L2, eps 1e-6) as used at src/train.py:64."""
    total = torch.sqrt(sum((g.detach().double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


def r2plus1d_loss_and_grads(x, y, params: Dict[str, torch.Tensor], bufs, layer_sizes, alpha,
                            loss: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], tap=None, force=None):
    """One forward+loss+backward of the oracle classifier.  Returns (logits, loss, grads dict).  tap / force: the
    LeakyReLU diagnostics of r2plus1d._leaky."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    logits = r2plus1d.classifier_forward(x, leaves, bufs, layer_sizes, alpha, training=True, tap=tap, force=force)
    L = loss(logits, y)
    L.backward()
    return logits.detach(), L.detach(), {k: v.grad for k, v in leaves.items()}


def dp_mean_grads(shards: List, params, bufs_per_rank, layer_sizes, alpha, loss):
    """Derived DP oracle: per-rank local BN stats, gradient MEAN over ranks."""
    acc = None
    outs = []
    for (x, y), bufs in zip(shards, bufs_per_rank):
        logits, L, g = r2plus1d_loss_and_grads(x, y, params, bufs, layer_sizes, alpha, loss)
        outs.append((logits, L))
        if acc is None:
            acc = {k: v.clone() for k, v in g.items()}
        else:
            for k in acc:
                acc[k] += g[k]
    for k in acc:
        acc[k] /= len(shards)
    return outs, acc

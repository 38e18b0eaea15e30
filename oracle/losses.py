"""Oracle: fp32 CPU restatement of the reference losses (test infrastructure).

Reference anchors:
  * FocalLoss         src/loss.py:14-34   sum_i w[y_i] * (1 - p_i)^gamma * ce_i,  p_i = exp(-ce_i)   (SUM reduction)
  * LDAMLoss          src/loss.py:37-69   margin m_j = max_m * n_j^-1/4 / max_k n_k^-1/4 subtracted from the
                                          target logit, logits scaled by s, class-weighted MEAN cross entropy
  * CELoss            src/loss.py:71-81   class-weighted cross entropy, SUM reduction
  * GradientBlending  src/GradientBlending.py:45-50  scale*(w_v L(vis) + w_t L(ts) + w_m L(fused))
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch


def log_softmax_rows(x: torch.Tensor) -> torch.Tensor:
    m = x.max(dim=1, keepdim=True).values
    z = x - m
    return z - torch.log(torch.exp(z).sum(dim=1, keepdim=True))


def focal_loss(logits: torch.Tensor, target: torch.Tensor, weight: torch.Tensor, gamma: float = 2.0) -> torch.Tensor:
    lsm = log_softmax_rows(logits)
    ce = -lsm.gather(1, target.view(-1, 1)).squeeze(1)      # F.cross_entropy(reduction='none')  loss.py:34
    p = torch.exp(-ce)                                       # loss.py:26
    a = weight.to(logits.dtype)[target]                      # loss.py:32
    return (a * (1.0 - p) ** gamma * ce).sum()               # loss.py:27-28


def ldam_margins(cls_num_list: Sequence[float], max_m: float = 0.5) -> torch.Tensor:
    m = 1.0 / np.sqrt(np.sqrt(np.asarray(cls_num_list, dtype=np.float64)))   # loss.py:53
    m = m * (max_m / np.max(m))                                              # loss.py:54
    return torch.tensor(m, dtype=torch.float32)                              # loss.py:55 (FloatTensor)


def ldam_loss(logits: torch.Tensor, target: torch.Tensor, m_list: torch.Tensor,
              weight: Optional[torch.Tensor], s: float = 30.0) -> torch.Tensor:
    onehot = torch.zeros_like(logits, dtype=torch.bool)
    onehot.scatter_(1, target.view(-1, 1), True)                             # loss.py:59-60
    batch_m = m_list[target].view(-1, 1)                                     # loss.py:62-64
    z = s * torch.where(onehot, logits - batch_m, logits)                    # loss.py:65-69
    lsm = log_softmax_rows(z)
    nll = -lsm.gather(1, target.view(-1, 1)).squeeze(1)
    if weight is None:
        return nll.mean()
    w = weight.to(logits.dtype)[target]
    return (w * nll).sum() / w.sum()                                         # F.cross_entropy weighted mean


def ce_loss(logits: torch.Tensor, target: torch.Tensor, weight: Optional[torch.Tensor]) -> torch.Tensor:
    lsm = log_softmax_rows(logits)
    nll = -lsm.gather(1, target.view(-1, 1)).squeeze(1)
    if weight is not None:
        nll = nll * weight.to(logits.dtype)[target]
    return nll.sum()                                                          # loss.py:81 reduction='sum'


def gradient_blending(loss_multi, loss_vis, loss_ts, w_vis: float, w_ts: float, w_multi: float, scale: float = 1.0):
    # GradientBlending.py:45-50 (argument order there: fused, vis, ts)
    return (loss_vis * scale) * w_vis + (loss_ts * scale) * w_ts + (loss_multi * scale) * w_multi


def drw_weights(epoch: int, num_epoch: int, betas: Sequence[float], cls_num_list: Sequence[int]) -> np.ndarray:
    """Deferred re-weighting schedule, src/train.py:318-329 (returns the fp32 values the
    reference puts in a FloatTensor)."""
    idx = epoch // int(num_epoch / len(betas))
    if idx >= len(betas):
        idx = len(betas) - 1
    beta = betas[idx]
    effective_num = 1.0 - np.power(beta, cls_num_list)
    w = (1.0 - beta) / np.array(effective_num)
    w = w / np.sum(w) * len(cls_num_list)
    return w.astype(np.float32)

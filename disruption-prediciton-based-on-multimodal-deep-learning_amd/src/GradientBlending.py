"""Gradient-Blending loss -- MI355X-native mirror of the reference's ``src/GradientBlending.py`` (the loss module).

``GradientBlending.forward(vis_ts_out, vis_out, ts_out, target)`` = scale*(w_vis*L(vis) + w_ts*L(ts) + w_multi*L(fused))
(reference GradientBlending.py:20-50).  Each of the three losses is one fused HIP softmax-loss launch
(``src.loss``); the weighted sum stays on the device.  The adaptive weight-estimation loops
(``GB_estimate`` / ``train_GB_dynamic``, reference :52-114, :310-446) are the next scope row and are
provided in their plain form: ``train_GB`` trains with fixed weights through this package's ``train_per_epoch``.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

from .train import train_per_epoch, valid_per_epoch


class GradientBlending(nn.Module):
    def __init__(self, loss_vis: nn.Module, loss_ts: nn.Module, loss_vis_ts: nn.Module, vis_weight: float = 0.0,
                 ts_weight: float = 0.0, vis_ts_weight: float = 1.0, loss_scale: float = 1.0):
        super().__init__()
        self.loss_vis = loss_vis
        self.loss_ts = loss_ts
        self.loss_vis_ts = loss_vis_ts
        self.vis_weight = vis_weight
        self.ts_weight = ts_weight
        self.vis_ts_weight = vis_ts_weight
        self.loss_scale = loss_scale

    def update_weights(self, ws: Dict):
        self.vis_weight = ws['video']
        self.ts_weight = ws['0D']
        self.vis_ts_weight = ws['multi']

    def forward(self, vis_ts_out: torch.Tensor, vis_out: torch.Tensor, ts_out: torch.Tensor, target: torch.Tensor):
        loss_vis = self.loss_vis(vis_out, target) * self.loss_scale
        loss_ts = self.loss_ts(ts_out, target) * self.loss_scale
        loss_vis_ts = self.loss_vis_ts(vis_ts_out, target) * self.loss_scale
        return loss_vis * self.vis_weight + loss_ts * self.ts_weight + loss_vis_ts * self.vis_ts_weight


def train_GB(train_loader, valid_loader, model, optimizer, scheduler, loss_fn: GradientBlending, device: str = "cpu",
             num_epoch: int = 64, verbose: Optional[int] = 8, save_best_dir: str = "./weights/best.pt",
             save_last_dir: str = "./weights/last.pt", exp_dir: Optional[str] = None, max_norm_grad: Optional[float] = None,
             **_ignored):
    """Fixed-weight gradient-blending training (reference GradientBlending.py:165-308): the epoch loop of
    ``src.train`` with model_type "multi-GB" (the model returns fused, vision and 0D logits)."""
    hist = {k: [] for k in ("train_loss", "train_acc", "train_f1", "valid_loss", "valid_acc", "valid_f1")}
    best_f1 = 0.0
    for epoch in range(num_epoch):
        if hasattr(model, "update_use_stream"):
            model.update_use_stream("multi-GB")
        tl, ta, tf = train_per_epoch(train_loader, model, optimizer, scheduler, loss_fn, device, max_norm_grad, "multi-GB")
        vl, va, vf = valid_per_epoch(valid_loader, model, optimizer, loss_fn, device, "multi-GB")
        for k, v in zip(hist, (tl, ta, tf, vl, va, vf)):
            hist[k].append(v)
        if verbose and epoch % verbose == 0:
            print("epoch : {}, train loss : {:.3f}, valid loss : {:.3f}, train f1 : {:.3f}, valid f1 : {:.3f}".format(
                epoch + 1, tl, vl, tf, vf))
        torch.save(model.state_dict(), save_last_dir)
        if vf > best_f1:
            best_f1 = vf
            torch.save(model.state_dict(), save_best_dir)
    return (hist["train_loss"], hist["train_acc"], hist["train_f1"], hist["valid_loss"], hist["valid_acc"], hist["valid_f1"])

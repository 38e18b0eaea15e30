"""Gradient-Blending loss -- MI355X-native mirror of the reference's ``src/GradientBlending.py`` (the loss module).

``GradientBlending.forward(vis_ts_out, vis_out, ts_out, target)`` = scale*(w_vis*L(vis) + w_ts*L(ts) + w_multi*L(fused))
(reference GradientBlending.py:20-50).  Each of the three losses is one fused HIP softmax-loss launch
(``src.loss``); the weighted sum stays on the device.  ``train_GB`` (reference :165-308) trains with fixed weights;
``GB_estimate`` (:52-114) and ``train_GB_dynamic`` (:310-446) are the adaptive weight estimation around it -- host logic
over this package's ``train_per_epoch`` / ``valid_per_epoch``, reproducing the reference's behaviour literally (see the
notes in their docstrings) with the evident intent available behind ``literal=False``.
"""
import os
from typing import Dict, Optional

import numpy as np

import torch
import torch.nn as nn

from .train import train_per_epoch, valid_per_epoch
from .utils.metrics import macro_f1


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


def evaluate_GB(test_loader, model, optimizer=None, device: Optional[str] = "cpu", threshold: float = 0.5):
    """Macro-F1 of the fused, the vision and the 0D head of a ``*_GB`` model over a loader (reference
    GradientBlending.py:116-163; used by its ``train_GB`` at :254-262 to monitor the three streams).

    The reference takes ``softmax(out).max(1)[1]`` -- the arg-max index, 0 or 1 -- and compares THAT with ``threshold``, so
    the prediction is the arg-max for any threshold in [0, 1) (softmax is monotone: arg-max of the logits).  Here the three
    arg-maxes stay on the device in per-loader buffers and are read back once at the end instead of three ``.cpu()``
    round trips per batch; ``optimizer`` is accepted for signature parity (the reference calls ``zero_grad`` under
    ``no_grad``, which changes nothing that is measured)."""
    if device is None:
        device = torch.device("cuda:0")
    model.to(device)
    model.eval()
    preds, labels = [], []
    thr = float(threshold)
    with torch.no_grad():
        for data, target in test_loader:
            out, out_vis, out_ts = model(data['video'].to(device), data['0D'].to(device))
            idx = torch.stack([out.argmax(1), out_vis.argmax(1), out_ts.argmax(1)], 0)        # (3, B) int64 on the device
            preds.append(idx.to(torch.float32) > thr)
            labels.append(target.to(device).reshape(-1))
    if not preds:
        return 0.0, 0.0, 0.0
    p = torch.cat(preds, 1).cpu().numpy().astype(np.float64)                                   # ONE read-back
    y = torch.cat(labels, 0).cpu().numpy().astype(np.float64)
    return macro_f1(y, p[0]), macro_f1(y, p[1]), macro_f1(y, p[2])


def train_GB(train_loader, valid_loader, model, optimizer, scheduler, loss_fn: GradientBlending, device: str = "cpu",
             num_epoch: int = 64, verbose: Optional[int] = 8, save_best_dir: str = "./weights/best.pt",
             save_last_dir: str = "./weights/last.pt", exp_dir: Optional[str] = None, max_norm_grad: Optional[float] = None,
             criteria: str = "f1_score", test_for_check_per_epoch=None):
    """Fixed-weight gradient-blending training (reference GradientBlending.py:165-308): the epoch loop of
    ``src.train`` with model_type "multi-GB" (the model returns fused, vision and 0D logits); the best checkpoint is chosen
    by ``criteria`` ("f1_score", "acc" or "loss") exactly as at :276-296.  After every epoch the three per-stream macro-F1
    scores of ``evaluate_GB`` over the training and the validation loader are taken as the reference does (:254-262; printed
    with the epoch report) and kept in ``train_GB.stream_f1`` = {"train": [(fusion, video, 0D), ...], "valid": [...]} of
    the last call -- the return value stays the reference's six lists.  TensorBoard logging and the per-epoch evaluation
    figure (:201-205, :240-251) are presentation and are left out (``test_for_check_per_epoch`` is accepted and ignored)."""
    hist = {k: [] for k in ("train_loss", "train_acc", "train_f1", "valid_loss", "valid_acc", "valid_f1")}
    train_GB.stream_f1 = {"train": [], "valid": []}
    best_acc, best_epoch, best_f1, best_loss = 0, 0, 0, float("inf")
    if exp_dir and not os.path.isdir(exp_dir):
        os.mkdir(exp_dir)
    for epoch in range(num_epoch):
        if hasattr(model, "update_use_stream"):
            model.update_use_stream("multi-GB")
        tl, ta, tf = train_per_epoch(train_loader, model, optimizer, scheduler, loss_fn, device, max_norm_grad, "multi-GB")
        vl, va, vf = valid_per_epoch(valid_loader, model, optimizer, loss_fn, device, "multi-GB")
        for k, v in zip(hist, (tl, ta, tf, vl, va, vf)):
            hist[k].append(v)
        f1_tr = evaluate_GB(train_loader, model, optimizer, device, 0.5)
        f1_va = evaluate_GB(valid_loader, model, optimizer, device, 0.5)
        train_GB.stream_f1["train"].append(f1_tr); train_GB.stream_f1["valid"].append(f1_va)
        if verbose and epoch % verbose == 0:
            print("# epoch: {}, train loss: {:.3f}, valid loss: {:.3f}".format(epoch + 1, tl, vl))
            print("# train, fusion: {:.3f}, video: {:.3f}, 0D : {:.3f}".format(*f1_tr))
            print("# valid, fusion: {:.3f}, video: {:.3f}, 0D : {:.3f}".format(*f1_va))
        better = ((criteria == "acc" and best_acc < va) or (criteria == "f1_score" and best_f1 < vf)
                  or (criteria == "loss" and best_loss > vl))
        if better:
            best_acc, best_f1, best_loss, best_epoch = va, vf, vl, epoch
            torch.save(model.state_dict(), save_best_dir)
        torch.save(model.state_dict(), save_last_dir)
    print("(Report) training process finished, best loss : {:.3f} and best acc : {:.3f}, best f1 : {:.3f}, best epoch : {}".format(
        best_loss, best_acc, best_f1, best_epoch))
    return (hist["train_loss"], hist["train_acc"], hist["train_f1"], hist["valid_loss"], hist["valid_acc"], hist["valid_f1"])


def GB_estimate(n_epochs: int, train_loader, valid_loader, multi_save_dir: str, multi_model, optimizer, scheduler, loss_fn,
                device: str = "cpu", max_norm_grad: Optional[float] = None, literal: bool = True) -> Dict[str, float]:
    """Gradient-Blending weights from the overfitting-to-generalisation ratio of each stream (reference :52-114): for the
    tasks "video", "0D", "multi": reload ``multi_save_dir``, train ``n_epochs`` with only that stream active, and set
    w = G / (O_f - O_i)^2 with O = valid - train loss (first / last epoch) and G = last - first valid loss; weights are
    normalised to sum 1.

    literal=True reproduces the reference exactly: its loss lists are NOT reset between tasks (:65-66 are outside the task
    loop), so O_i and the first valid loss always come from the first epoch of the "video" task, and the optimizer state
    is carried from task to task.  literal=False resets the lists per task (the evident intent of the formula)."""
    train_loss_list, valid_loss_list, w_list = [], [], []
    tasks = ["video", "0D", "multi"]
    for task in tasks:
        multi_model.load_state_dict(torch.load(multi_save_dir, weights_only=True))
        multi_model.update_use_stream(task)
        if not literal:
            train_loss_list, valid_loss_list = [], []
        for _ in range(n_epochs):
            train_loss, _, _ = train_per_epoch(train_loader, multi_model, optimizer, scheduler, loss_fn, device, max_norm_grad, "multi")
            valid_loss, _, _ = valid_per_epoch(valid_loader, multi_model, optimizer, loss_fn, device, "multi")
            train_loss_list.append(train_loss)
            valid_loss_list.append(valid_loss)
        Oi = valid_loss_list[0] - train_loss_list[0]
        Of = valid_loss_list[-1] - train_loss_list[-1]
        G = valid_loss_list[-1] - valid_loss_list[0]
        w_list.append(G / (Of - Oi) ** 2)
    w = np.array(w_list) / np.sum(w_list)
    return {key: wi for key, wi in zip(tasks, w)}


def train_GB_dynamic(train_loader, valid_loader, model, optimizer, scheduler, loss_GB: GradientBlending, loss_unimodal,
                     device: str = "cpu", num_epoch: int = 64, epoch_per_GB_estimate: int = 16, num_epoch_GB_estimate: int = 4,
                     verbose: Optional[int] = 8, save_best_dir: str = "./weights/best.pt",
                     save_last_dir: str = "./weights/last.pt", exp_dir: Optional[str] = None,
                     max_norm_grad: Optional[float] = None, criteria: str = "f1_score", test_for_check_per_epoch=None,
                     literal: bool = True):
    """Gradient-Blending training with periodic re-estimation of the weights (reference :310-446).

    literal=True keeps the reference's schedule test ``epoch % epoch_per_GB_estimate and epoch != 0`` (:411), which is true
    for every epoch that is NOT a multiple of the period; literal=False re-estimates on the multiples (the evident intent).
    TensorBoard logging and the per-epoch evaluation figure (:349-353, :403-409) are presentation and are left out."""
    model_type = "multi-GB"
    hist = {k: [] for k in ("train_loss", "train_acc", "train_f1", "valid_loss", "valid_acc", "valid_f1")}
    best_acc, best_epoch, best_f1, best_loss = 0, 0, 0, float("inf")
    if exp_dir and not os.path.isdir(exp_dir):
        os.mkdir(exp_dir)
    for epoch in range(num_epoch):
        model.update_use_stream("multi-GB")
        tl, ta, tf = train_per_epoch(train_loader, model, optimizer, scheduler, loss_GB, device, max_norm_grad, model_type)
        vl, va, vf = valid_per_epoch(valid_loader, model, optimizer, loss_GB, device, model_type)
        for k, v in zip(hist, (tl, ta, tf, vl, va, vf)):
            hist[k].append(v)
        if verbose and epoch % verbose == 0:
            print("epoch : {}, train loss : {:.3f}, valid loss : {:.3f}, train acc : {:.3f}, valid acc : {:.3f}, train f1 : {:.3f}, "
                  "valid f1 : {:.3f}".format(epoch + 1, tl, vl, ta, va, tf, vf))
        due = (epoch % epoch_per_GB_estimate and epoch != 0) if literal else (epoch % epoch_per_GB_estimate == 0 and epoch != 0)
        if due:
            ws = GB_estimate(num_epoch_GB_estimate, train_loader, valid_loader, save_last_dir, model, optimizer, scheduler,
                             loss_unimodal, device, max_norm_grad, literal=literal)
            loss_GB.update_weights(ws)
        better = ((criteria == "acc" and best_acc < va) or (criteria == "f1_score" and best_f1 < vf)
                  or (criteria == "loss" and best_loss > vl))
        if better:
            best_acc, best_f1, best_loss, best_epoch = va, vf, vl, epoch
            torch.save(model.state_dict(), save_best_dir)
        torch.save(model.state_dict(), save_last_dir)
    model.update_use_stream("multi-GB")
    print("(Report) training process finished, best loss : {:.3f} and best acc : {:.3f}, best f1 : {:.3f}, best epoch : {}".format(
        best_loss, best_acc, best_f1, best_epoch))
    return (hist["train_loss"], hist["train_acc"], hist["train_f1"], hist["valid_loss"], hist["valid_acc"], hist["valid_f1"])

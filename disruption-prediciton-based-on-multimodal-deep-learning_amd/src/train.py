"""Epoch loops -- MI355X-native mirror of the reference's ``src/train.py`` (same names and signatures).

``train_per_epoch`` keeps the reference's step semantics (src/train.py:17-93: zero_grad -> forward -> loss ->
finite check -> backward -> clip_grad_norm_ -> optimizer.step -> argmax bookkeeping -> macro-F1) but removes the
per-step host round trips that are not control flow: the loss sum, the correct-count and the predictions stay on
the device and are read back once per epoch; ``pred`` comes from the fused loss kernel when the loss module
provides it (``last_pred``) instead of a second softmax pass.  The finite-loss guard (:56-58) is control flow in
the reference and stays a (single) host read per step.

Unlike the reference this module does NOT switch on autograd anomaly mode at import (src/train.py:15): it only
slows backward and the kernels here have no autograd graph inside to inspect.
"""
from typing import List, Literal, Optional, Union

import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from .loss import FocalLoss, LDAMLoss
from .utils.metrics import macro_f1

try:  # optional, exactly as optional as in the reference's environment
    from torch.utils.tensorboard import SummaryWriter  # type: ignore
except Exception:  # pragma: no cover
    class SummaryWriter:  # minimal stand-in: logging is not part of the hot path
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def add_figure(self, *a, **k):
            pass

        def close(self):
            pass

try:
    from tqdm.auto import tqdm
except Exception:  # pragma: no cover
    def tqdm(it, **k):
        return it


def _forward(model, data, device, model_type):
    if model_type == "single":
        return model(data.to(device)), None, None
    if model_type == "multi":
        return model(data['video'].to(device), data['0D'].to(device)), None, None
    out, out_vis, out_ts = model(data['video'].to(device), data['0D'].to(device))
    return out, out_vis, out_ts


# MD_GRAPH_STEP=1: forward + loss + backward of every full-size training batch replayed from one HIP graph
# (src/utils/graphed.py; the composable models -- SlowFast, ViViT, the 0D encoders, the fusion models -- are launch-bound:
# cfg5 16 ms eager, 7.1 ms replayed).  The optimizer step, the bookkeeping and odd-sized batches stay eager.
_GRAPH_STEPS = os.environ.get("MD_GRAPH_STEP") == "1"


def _loss_signature(loss_fn):
    """Everything of the loss modules a captured step bakes in: numbers by value, tensors by address (a replay reads their
    current contents)."""
    sig = []
    for mod in loss_fn.modules():
        for k, v in sorted(vars(mod).items()):
            if k.startswith("_") or k == "training" or k == "last_pred":
                continue
            if isinstance(v, (bool, int, float, str)) or v is None:
                sig.append((k, v))
            elif isinstance(v, torch.Tensor):
                sig.append((k, v.data_ptr(), tuple(v.shape)))
            elif isinstance(v, (list, tuple)) and all(isinstance(e, (bool, int, float)) for e in v):
                sig.append((k, tuple(v)))
        for k, v in list(mod._buffers.items()) + list(mod._parameters.items()):
            if v is not None:
                sig.append((k, v.data_ptr(), tuple(v.shape)))
    return tuple(sig)


def _graphed_step(model, loss_fn, inputs, tgt, model_type):
    """The cached GraphedStep of (model, loss configuration, batch shapes), captured on first use; None when capture is not
    possible (then the loop stays eager for good and says so once)."""
    from .utils.graphed import GraphedStep
    key = (model_type, getattr(model, "use_stream", None), tuple(tuple(t.shape) for t in inputs), tuple(tgt.shape), tgt.dtype,
           id(loss_fn), _loss_signature(loss_fn))
    slot = model.__dict__.get("_md_graphed")
    if slot is not None and slot[0] == key:
        return slot[1]
    if slot is not None and slot[1] is None and slot[0][:2] == key[:2] and slot[0][5] == key[5]:
        return None                                           # capture already failed for this model / loss pair
    model.__dict__.pop("_md_graphed", None)                   # (frees the previous graph and its memory pool)
    slot = None
    try:
        gs = GraphedStep(model, loss_fn, inputs, tgt, keep_buffers=True)
    except RuntimeError as e:
        print("train_per_epoch | MD_GRAPH_STEP: capture refused, training eagerly (%s)" % str(e).split("\n")[0][:200])
        gs = None
    model.__dict__["_md_graphed"] = (key, gs)
    return gs


# One C call per training step where the (model, loss, optimizer) triple allows it (src/_step.py: R2Plus1DClassifier + Focal / LDAM /
# CE + ClipAdamW); MD_FUSED_STEP=0 keeps the composed step.
_FUSED_STEPS = os.environ.get("MD_FUSED_STEP", "1") != "0"


def _fused_step(model, loss_fn, optimizer):
    from . import _step
    if not _step.applicable(model, loss_fn, optimizer):        # looked at once per epoch (hooks, parameter groups ... can change)
        return None
    key = (id(loss_fn), id(optimizer))
    slot = model.__dict__.get("_md_fused")
    if slot is not None and slot[0] == key:
        return slot[1]
    fs = _step.FusedTrainStep(model, loss_fn, optimizer)
    model.__dict__["_md_fused"] = (key, fs)
    return fs


def _pred_of(loss_fn, output):
    """argmax softmax(output) (src/train.py:70).  The fused loss kernel already produced it for `output`."""
    p = getattr(loss_fn, "last_pred", None)
    if p is None and hasattr(loss_fn, "loss_vis_ts"):      # GradientBlending: bookkeeping uses the fused logits
        p = getattr(loss_fn.loss_vis_ts, "last_pred", None)
    if p is not None and p.numel() == output.size(0):
        return p.view(-1, 1)
    return torch.nn.functional.softmax(output, dim=1).max(1, keepdim=True)[1]


def train_per_epoch(
        train_loader: DataLoader,
        model: torch.nn.Module,
        optimizer: torch.optim.Optimizer,
        scheduler: Optional[torch.optim.lr_scheduler._LRScheduler],
        loss_fn: torch.nn.Module,
        device: str = "cpu",
        max_norm_grad: Optional[float] = None,
        model_type: Literal["single", "multi", "multi-GB"] = "single",
):
    model.train()
    model.to(device)

    loss_sum = None
    correct = None
    total_pred, total_label = [], []
    total_size = 0

    graph_ok = _GRAPH_STEPS and torch.device(device).type == "cuda" and isinstance(loss_fn, torch.nn.Module)
    fused = None
    if _FUSED_STEPS and not graph_ok and model_type == "single" and torch.device(device).type == "cuda":
        fused = _fused_step(model, loss_fn, optimizer)
    fused_ok = []                # (index into total_pred, batch_idx, device flag) of the fused steps: read once, after the loop
    full_shape = None            # graph mode: the shape of the first batch is the one that is captured
    was_eager = True
    for batch_idx, (data, target) in enumerate(train_loader):
        tgt = target.to(device)
        gs = None
        if fused is not None:
            # forward + loss + backward + clip + update in one call; a non-finite loss leaves the parameters alone (device flag)
            loss, output, pred, ok = fused(data.to(device), tgt, max_norm=max_norm_grad)
            okb = ok > 0
            ld = torch.where(okb, loss, torch.zeros_like(loss))
            loss_sum = ld if loss_sum is None else loss_sum + ld
            c = (pred.eq(tgt.view_as(pred)) & okb).sum()
            correct = c if correct is None else correct + c
            fused_ok.append((len(total_pred), batch_idx, ok))
            total_pred.append(pred.view(-1, 1))
            total_label.append(tgt.view(-1, 1))
            total_size += pred.size(0)
            continue
        if graph_ok:
            inputs = [data.to(device)] if model_type == "single" else [data['video'].to(device), data['0D'].to(device)]
            shape = tuple(tuple(t.shape) for t in inputs)
            if full_shape is None:
                full_shape = shape
            if shape == full_shape:
                gs = _graphed_step(model, loss_fn, inputs, tgt, model_type)
        if gs is not None:
            if was_eager:
                gs.bind()
                was_eager = False
            outs, loss = gs(inputs, tgt)
            output = outs[0] if isinstance(outs, tuple) else outs
            if not torch.isfinite(loss):                      # (the replay already ran the backward: only the update is skipped)
                print("train_per_epoch | Warning : loss nan occurs at batch_idx : {}".format(batch_idx))
                continue
        else:
            was_eager = True
            optimizer.zero_grad()
            output, output_vis, output_ts = _forward(model, data, device, model_type)
            if model_type == 'multi-GB':
                loss = loss_fn(output, output_vis, output_ts, tgt)
            else:
                loss = loss_fn(output, tgt)

            if not torch.isfinite(loss):
                print("train_per_epoch | Warning : loss nan occurs at batch_idx : {}".format(batch_idx))
                continue
            loss.backward()

        if getattr(optimizer, "fused_clip", False):         # src.optim.ClipAdamW: clip + update in one pass
            optimizer.step(max_norm=max_norm_grad)
        else:
            if max_norm_grad:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm_grad)
            optimizer.step()

        ld = loss.detach()
        pred = _pred_of(loss_fn, output.detach())
        if gs is not None:                                    # static tensors of the graph: the next replay overwrites them
            ld, pred = ld.clone(), pred.clone()
        loss_sum = ld if loss_sum is None else loss_sum + ld
        c = pred.eq(tgt.view_as(pred)).sum()
        correct = c if correct is None else correct + c
        total_size += pred.size(0)
        total_pred.append(pred.view(-1, 1))
        total_label.append(tgt.view(-1, 1))

    if fused_ok:
        flags = torch.stack([f for _, _, f in fused_ok]).cpu().tolist()     # the epoch's one read of the finite-loss flags
        drop = set()
        for (slot, bidx, _), f in zip(fused_ok, flags):
            if f != 1.0:
                print("train_per_epoch | Warning : loss nan occurs at batch_idx : {}".format(bidx))
                drop.add(slot)
        total_pred = [t for i, t in enumerate(total_pred) if i not in drop]
        total_label = [t for i, t in enumerate(total_label) if i not in drop]
        total_size = sum(t.size(0) for t in total_pred)

    if scheduler:
        scheduler.step()

    if total_size > 0:
        preds = torch.concat(total_pred, dim=0).view(-1).cpu().numpy()
        labels = torch.concat(total_label, dim=0).view(-1).cpu().numpy()
        train_loss = float(loss_sum.item()) / total_size
        train_acc = int(correct.item()) / total_size
        train_f1 = macro_f1(labels, preds)
    else:
        train_loss, train_acc, train_f1 = 0, 0, 0
    return train_loss, train_acc, train_f1


def valid_per_epoch(
        valid_loader: DataLoader,
        model: torch.nn.Module,
        optimizer: torch.optim.Optimizer,
        loss_fn: torch.nn.Module,
        device: str = "cpu",
        model_type: Literal["single", "multi", "multi-GB"] = "single",
):
    model.eval()
    model.to(device)
    loss_sum = None
    correct = None
    total_pred, total_label = [], []
    total_size = 0
    graph_ok = _GRAPH_STEPS and torch.device(device).type == "cuda"
    full_shape = None            # graph mode: the first batch's shape is the one that is captured (the short last batch runs eagerly)
    for batch_idx, (data, target) in enumerate(valid_loader):
        with torch.no_grad():
            optimizer.zero_grad()
            out = None
            if graph_ok:
                inputs = [data.to(device)] if model_type == "single" else [data['video'].to(device), data['0D'].to(device)]
                shape = tuple(tuple(t.shape) for t in inputs)
                full_shape = shape if full_shape is None else full_shape
                if shape == full_shape:
                    from .utils.graphed import graphed_forward
                    out = graphed_forward(model, inputs, "_md_graphed_eval")
                    out = tuple(o.clone() for o in out) if isinstance(out, tuple) else out.clone()      # (static tensors of the graph)
            if out is None:
                output, output_vis, output_ts = _forward(model, data, device, model_type)
            elif model_type == "multi-GB":
                output, output_vis, output_ts = out
            else:
                output, output_vis, output_ts = out, None, None
            tgt = target.to(device)
            if model_type == 'multi-GB':
                loss = loss_fn(output, output_vis, output_ts, tgt)
            else:
                loss = loss_fn(output, tgt)
            ld = loss.detach()
            loss_sum = ld if loss_sum is None else loss_sum + ld
            pred = _pred_of(loss_fn, output)
            c = pred.eq(tgt.view_as(pred)).sum()
            correct = c if correct is None else correct + c
            total_size += pred.size(0)
            total_pred.append(pred.view(-1, 1))
            total_label.append(tgt.view(-1, 1))
    valid_loss = float(loss_sum.item()) / total_size
    valid_acc = int(correct.item()) / total_size
    preds = torch.concat(total_pred, dim=0).view(-1).cpu().numpy()
    labels = torch.concat(total_label, dim=0).view(-1).cpu().numpy()
    valid_f1 = macro_f1(labels, preds)
    return valid_loss, valid_acc, valid_f1


class _EarlyStop:
    """Validation-F1 early stopping with best-checkpoint save (reference src/utils/EarlyStopping.py:15-38)."""

    def __init__(self, path, patience, verbose, delta):
        self.path, self.patience, self.verbose, self.delta = path, patience, verbose, delta
        self.best, self.count, self.early_stop = None, 0, False

    def __call__(self, score, model):
        if self.best is None or score > self.best + self.delta:
            self.best, self.count = score, 0
            torch.save(model.state_dict(), self.path)
        else:
            self.count += 1
            if self.count >= self.patience:
                self.early_stop = True


def drw_class_weights(epoch: int, num_epoch: int, betas: List, cls_num_list: List) -> np.ndarray:
    """Deferred re-weighting schedule (reference src/train.py:318-329); fp32 values, bit-exact host arithmetic."""
    idx = epoch // int(num_epoch / len(betas))
    if idx >= len(betas):
        idx = len(betas) - 1
    beta = betas[idx]
    effective_num = 1.0 - np.power(beta, cls_num_list)
    per_cls_weights = (1.0 - beta) / np.array(effective_num)
    per_cls_weights = per_cls_weights / np.sum(per_cls_weights) * len(cls_num_list)
    return per_cls_weights.astype(np.float32)


def _run(train_loader, valid_loader, model, optimizer, scheduler, loss_fn, device, num_epoch, verbose, save_best_dir,
         save_last_dir, exp_dir, max_norm_grad, model_type, is_early_stopping, es_verbose, es_patience, es_delta,
         drw=None, desc="training process"):
    lists = [[] for _ in range(6)]
    best_f1, best_acc, best_epoch, best_loss = 0, 0, 0, float("inf")
    if exp_dir and not os.path.isdir(exp_dir):
        os.makedirs(exp_dir, exist_ok=True)
    writer = SummaryWriter(exp_dir) if exp_dir else None
    early = _EarlyStop(save_best_dir, es_patience, es_verbose, es_delta) if is_early_stopping else None
    for epoch in tqdm(range(num_epoch), desc=desc):
        if drw is not None:
            betas, cls_num_list = drw
            w = torch.from_numpy(drw_class_weights(epoch, num_epoch, betas, cls_num_list)).to(device)
            loss_fn.update_weight(w)
        tl, ta, tf = train_per_epoch(train_loader, model, optimizer, scheduler, loss_fn, device, max_norm_grad, model_type)
        vl, va, vf = valid_per_epoch(valid_loader, model, optimizer, loss_fn, device, model_type)
        for lst, v in zip(lists, (tl, ta, tf, vl, va, vf)):
            lst.append(v)
        if writer is not None:
            writer.add_scalar('Loss/train', tl, epoch); writer.add_scalar('Loss/valid', vl, epoch)
            writer.add_scalar('F1_score/train', tf, epoch); writer.add_scalar('F1_score/valid', vf, epoch)
        if verbose and epoch % verbose == 0:
            print("epoch : {}, train loss : {:.3f}, valid loss : {:.3f}, train f1 : {:.3f}, valid f1 : {:.3f}".format(
                epoch + 1, tl, vl, tf, vf))
        torch.save(model.state_dict(), save_last_dir)
        if best_f1 < vf:
            best_acc, best_f1, best_loss, best_epoch = va, vf, vl, epoch
            if early is None:
                torch.save(model.state_dict(), save_best_dir)
        if early is not None:
            early(vf, model)
            if early.early_stop:
                print("Early stopping | epoch : {}, best f1 score : {:.3f}".format(epoch, best_f1))
                break
    print("training process finished, best loss : {:.3f}, best acc : {:.3f}, best f1 : {:.3f}, best epoch : {}".format(
        best_loss, best_acc, best_f1, best_epoch))
    if writer:
        writer.close()
    tl_, ta_, tf_, vl_, va_, vf_ = lists
    return tl_, ta_, tf_, vl_, va_, vf_


def train(
        train_loader: DataLoader, valid_loader: DataLoader, model: torch.nn.Module, optimizer: torch.optim.Optimizer,
        scheduler: Optional[torch.optim.lr_scheduler._LRScheduler], loss_fn: torch.nn.Module, device: str = "cpu",
        num_epoch: int = 64, verbose: Optional[int] = 8, save_best_dir: str = "./weights/best.pt",
        save_last_dir: str = "./weights/last.pt", exp_dir: Optional[str] = None, max_norm_grad: Optional[float] = None,
        model_type: Literal["single", "multi", "multi-GB"] = "single", test_for_check_per_epoch: Optional[DataLoader] = None,
        is_early_stopping: bool = False, early_stopping_verbose: bool = True, early_stopping_patience: int = 12,
        early_stopping_delta: float = 1e-3,
):
    """Reference src/train.py:147-274 (figure logging via evaluate_tensorboard is out of scope)."""
    return _run(train_loader, valid_loader, model, optimizer, scheduler, loss_fn, device, num_epoch, verbose, save_best_dir,
                save_last_dir, exp_dir, max_norm_grad, model_type, is_early_stopping, early_stopping_verbose,
                early_stopping_patience, early_stopping_delta, None, "training process")


def train_DRW(
        train_loader: DataLoader, valid_loader: DataLoader, model: torch.nn.Module, optimizer: torch.optim.Optimizer,
        loss_fn: Union[LDAMLoss, FocalLoss], device: str = "cpu", num_epoch: int = 64, verbose: int = 1,
        save_best_dir: str = "./weights/best.pt", save_last_dir: str = "./weights/last.pt", exp_dir: str = './results',
        max_norm_grad: Optional[float] = None, cls_num_list: Optional[List] = None, betas: List = [0, 0.25, 0.75, 0.9],
        model_type: Literal['single', 'multi'] = 'single', test_for_check_per_epoch: Optional[DataLoader] = None,
        is_early_stopping: bool = False, early_stopping_verbose: bool = True, early_stopping_patience: int = 12,
        early_stopping_delta: float = 1e-3,
):
    """Reference src/train.py:277-422: per-epoch class re-weighting, no LR scheduler."""
    return _run(train_loader, valid_loader, model, optimizer, None, loss_fn, device, num_epoch, verbose, save_best_dir,
                save_last_dir, exp_dir, max_norm_grad, model_type, is_early_stopping, early_stopping_verbose,
                early_stopping_patience, early_stopping_delta, (betas, cls_num_list),
                "training process - Deferred Re-weighting")

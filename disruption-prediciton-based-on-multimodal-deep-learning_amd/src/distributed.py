"""Data-parallel loop -- MI355X-native counterpart of the reference's ``src/distributed.py``.

Same entry points (``set_random_seeds``, ``get_distributed_loader``, ``train_epoch_per_procs``, ``train_per_proc``,
``train_distributed``) with the INTENDED semantics of the reference sketch: one process per GPU, synchronous
gradient-mean data parallelism, one optimizer for the whole run.  The reference itself never exchanges gradients
(it runs the un-wrapped model, src/distributed.py:74) and rebuilds DDP/optimizer every epoch (:46-52); neither is
copied.

Exchange design for RCCL over xGMI (point-to-point links, small latency-bound messages):
  * the trunk backward produces ONE flat fp32 gradient buffer (``_plan.TrunkFunction``); after the backward of each
    residual stage its weight slice is all-reduced (AVG) asynchronously on RCCL's stream while the earlier stages'
    backward kernels keep the compute stream busy -- 5 messages of 0.1-3 MB instead of 201 tiny ones;
  * BatchNorm gammas/betas (one contiguous tail of the same buffer) go in one message; every other gradient (head,
    other encoders) lives in ONE pre-flattened bucket whose last element is the rank's "loss is NOT finite" indicator (0 / 1),
    so the non-finite-loss decision (reference src/train.py:56-58) rides on that all-reduce: collective, decided on the device
    (the update is applied iff the reduced indicator is exactly 0 -- a sum or average of zeros is exact for every world size,
    reduction order and algorithm, which an average of ones is not: 1/3 + 1/3 + 1/3 need not round to 1), and no rank skips a
    collective the others wait in; the step contains no host synchronisation;
  * BatchNorm statistics stay per rank (the reference has no SyncBN); parameters and buffers are broadcast from
    rank 0 once.
"""
from __future__ import annotations

import os
import random
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler


def set_random_seeds(random_seed: int = 42):
    torch.manual_seed(random_seed)
    np.random.seed(random_seed)
    random.seed(random_seed)


def get_distributed_loader(train_dataset: Dataset, valid_dataset: Dataset, num_replicas: int, rank: int, num_workers: int,
                           batch_size: int = 32):
    train_sampler = DistributedSampler(train_dataset, num_replicas=num_replicas, rank=rank, shuffle=True)
    valid_sampler = DistributedSampler(valid_dataset, num_replicas=num_replicas, rank=rank, shuffle=False)
    train_loader = DataLoader(train_dataset, batch_size, sampler=train_sampler, num_workers=num_workers, pin_memory=True)
    valid_loader = DataLoader(valid_dataset, batch_size, sampler=valid_sampler, num_workers=num_workers, pin_memory=True)
    return train_loader, valid_loader


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Parameters and buffers from rank `src` (what DDP's constructor does, reference :46)."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


class GradAllReducer:
    """Averages gradients across ranks.  With an R(2+1)D trunk inside `module`, the trunk's weight gradients are
    reduced stage by stage during backward (see module docstring).  Every other parameter's gradient lives in ONE
    pre-flattened bucket (``p.grad`` are views into it, so autograd accumulates straight into the bucket: no cat, no copy back)
    whose last element carries the rank's "loss is not finite" indicator; one all-reduce(AVG) after backward averages the
    gradients and -- since a sum or average of 0/1 indicators is exactly 0 only if every rank said 0 -- decides the reference's
    non-finite skip (src/train.py:56-58) collectively and on the device.

    Deviation from ``optimizer.zero_grad()`` + AdamW (documented, tests/test_dp_cpu.py): a parameter outside the trunk that
    takes no part in a step (an unused encoder of a fusion model) keeps a ZERO gradient slice here instead of ``None``, so AdamW
    still applies weight decay and moment decay to it; the reference's loop would skip it.  Freeze such parameters
    (``requires_grad_(False)``) to keep them out of the bucket."""

    def __init__(self, module: torch.nn.Module, group=None):
        self.module = module
        self.group = group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.pending: List = []
        self.log_messages = False       # bench / tests: record (tag, bytes) of every collective of a step in self.messages
        self.messages: List = []
        self.trunk = None
        self._trunk_param_ids = set()
        for m in module.modules():
            if hasattr(m, "grad_segment_hook") and hasattr(m, "unit_modules"):
                self.trunk = m
                m.grad_segment_hook = self._segment_hook
                for u in m.unit_modules():
                    for p in (u.conv.weight, u.bn.weight, u.bn.bias):
                        self._trunk_param_ids.add(id(p))
                break
        self._stage_slices = None
        self.rest = [p for p in module.parameters() if p.requires_grad and id(p) not in self._trunk_param_ids]
        dev = next(module.parameters()).device
        n = sum(p.numel() for p in self.rest)
        self.flat_rest = torch.zeros(n + 1, device=dev, dtype=torch.float32)      # [gradients .. | finite flag]
        self.flag = self.flat_rest[n:]                                            # 1-element view
        self._views, o = [], 0
        for p in self.rest:
            self._views.append(self.flat_rest[o:o + p.numel()].view_as(p)); o += p.numel()
        # does the collective library average natively?  probed ONCE, synchronously (an asynchronous failure could not be caught)
        self._native_avg = False
        if self.backend == "nccl":
            try:
                probe = torch.zeros(1, device=dev)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=group)
                self._native_avg = bool(probe.item() == 0.0)
            except Exception:
                self._native_avg = False

    # -- helpers
    def _avg(self, t: torch.Tensor, async_op: bool, tag: str = ""):
        if self.log_messages:
            self.messages.append((tag, t.numel() * t.element_size()))
        if self._native_avg:
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return (w, t)
        t.div_(self.world)
        return None

    def _wait(self, h):
        if isinstance(h, tuple):
            h[0].wait(); h[1].div_(self.world)
        elif h is not None:
            h.wait()

    def zero_grad(self) -> None:
        """Replaces optimizer.zero_grad() in the data-parallel step: trunk parameters get fresh gradient views from the
        executor's flat buffer every backward (grad = None); every other parameter's .grad is (re)pointed at its slice of the
        zeroed bucket."""
        for p in self.module.parameters():
            if id(p) in self._trunk_param_ids:
                p.grad = None
        self.flat_rest.zero_()
        for p, v in zip(self.rest, self._views):
            p.grad = v

    def _build_stage_slices(self, grads):
        """weight-gradient range (in floats) of each stage inside the flat buffer; stage of unit from the plan order."""
        units = self.trunk.unit_modules()
        n = len(units)
        stage_of = [0, 0]
        for si, layer in enumerate((self.trunk.conv2, self.trunk.conv3, self.trunk.conv4, self.trunk.conv5)):
            for blk in [layer.block1] + list(layer.blocks):
                stage_of += [si + 1] * (6 if blk.downsample else 4)
        assert len(stage_of) == n
        offs, o = [], 0
        for gten in grads[:n]:
            offs.append((o, o + gten.numel())); o += gten.numel()
        sl = {}
        for st in range(5):
            idx = [i for i in range(n) if stage_of[i] == st]
            sl[st] = (offs[idx[0]][0], offs[idx[-1]][1])
        self._w_end = o
        self._stage_slices = sl

    # -- called from TrunkFunction.backward after the backward of stage `st` (4 .. 0)
    def _segment_hook(self, st: int, flat: torch.Tensor, grads, stream=None) -> None:
        """`stream`: the stream that produces this stage's weight gradients when it is not the current one (the plan's
        side stream); the all-reduce is queued behind it."""
        if self._stage_slices is None:
            self._build_stage_slices(grads)
        a, b = self._stage_slices[st]
        if stream is not None:
            with torch.cuda.stream(stream):
                self.pending.append(self._avg(flat[a:b], async_op=True, tag="trunk.stage%d.weights" % st))
        else:
            self.pending.append(self._avg(flat[a:b], async_op=True, tag="trunk.stage%d.weights" % st))
        if st == 0:
            self.pending.append(self._avg(flat[self._w_end:], async_op=True, tag="trunk.bn"))   # all gammas and betas
            for h in self.pending:
                self._wait(h)
            self.pending = []

    def reduce_rest(self, finite: Optional[torch.Tensor] = None) -> torch.Tensor:
        """After loss.backward(): ONE all-reduce over the pre-flattened bucket of every gradient the trunk hook does not cover
        (head, other encoders) plus the indicator (``finite``: this rank's 0/1 device scalar, 1 = finite; default 1).  Returns
        ``ok`` (device, 1 element): exactly 1.0 iff the loss was finite on every rank, else 0.0."""
        for p, v in zip(self.rest, self._views):
            if p.grad is None:                       # parameter took no part in this step: contributes zeros
                continue
            if p.grad.data_ptr() != v.data_ptr():    # someone replaced .grad (e.g. optimizer.zero_grad(set_to_none=True)): fold it in
                v.copy_(p.grad); p.grad = v
        if finite is None:
            self.flag.fill_(0.0)
        else:
            self.flag.copy_(1.0 - finite.reshape(1).to(torch.float32))      # "bad" indicator: 0 = finite
        self._avg(self.flat_rest, async_op=False, tag="rest+flag")
        return (self.flag == 0).to(torch.float32)    # (the bucket is zeroed again by the next step's zero_grad)


def all_ranks_finite(loss: torch.Tensor, group=None) -> bool:
    """Collective form of the reference's `if not torch.isfinite(loss): continue` (src/train.py:56-58) WITH a host
    synchronisation; the data-parallel step uses the device-side flag of GradAllReducer.reduce_rest instead."""
    flag = torch.isfinite(loss.detach()).to(torch.float32).reshape(1)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() > 0)


def dp_train_step(model, reducer: GradAllReducer, optimizer, loss_fn, data, target, max_norm_grad: Optional[float] = None):
    """One synchronous data-parallel optimisation step.  Returns (local detached loss, detached logits, ok) where ``ok`` is
    the all-rank finite flag as a DEVICE tensor (1.0 = the step was applied on every rank; anything else = skipped on every
    rank) -- nothing in here synchronises the host when the optimizer is src.optim.ClipAdamW.  Backward always runs (a rank
    with a non-finite loss contributes garbage gradients that no rank applies), so no rank can skip a collective the others
    wait in."""
    reducer.zero_grad()
    output = model(data)
    loss = loss_fn(output, target)
    finite = torch.isfinite(loss.detach()).to(torch.float32)
    loss.backward()
    ok = reducer.reduce_rest(finite)
    if getattr(optimizer, "fused_clip", False):             # src.optim.ClipAdamW: clip + update in one pass, device-side skip
        optimizer.step(max_norm=max_norm_grad, ok=ok)
    else:
        if bool(ok.item() == 1.0):                          # torch optimizers: host decision (one sync per step)
            if max_norm_grad:
                torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm_grad)
            optimizer.step()
    return loss.detach(), output.detach(), ok


def _per_sample(loss: torch.Tensor, loss_fn, n: int) -> torch.Tensor:
    """Loss per sample for the epoch bookkeeping: the package's Focal / CE losses SUM over the batch, LDAM and torch's default
    criteria average (src/loss.py:28,69,81 -- SURVEY Q3)."""
    reduction = getattr(loss_fn, "reduction", None)
    if reduction is None:
        reduction = "mean" if type(loss_fn).__name__ == "LDAMLoss" else ("sum" if type(loss_fn).__name__ in ("FocalLoss", "CELoss") else "mean")
    return loss / n if reduction == "sum" else loss


def train_epoch_per_procs(rank: int, world_size: int, batch_size: Optional[int], model: torch.nn.Module,
                          train_dataset: Dataset, valid_dataset: Dataset, random_seed: int = 42, resume: bool = True,
                          loss_fn=None, model_filepath: str = "./weights/distributed.pt", optimizer=None, scheduler=None,
                          reducer: Optional[GradAllReducer] = None, epoch: int = 0):
    """One epoch on this rank (reference :29-111).  Returns (train_loss, train_acc, valid_loss, valid_acc): per-sample loss
    and accuracy averaged over the batches of every rank.  Rank r sees samples r::world of the epoch's permutation
    (DistributedSampler with set_epoch, which the reference forgot -- SURVEY Q2)."""
    device = torch.device("cuda:{}".format(rank)) if torch.cuda.is_available() else torch.device("cpu")
    if loss_fn is None:
        from .loss import CELoss
        loss_fn = CELoss(weight=None)
    train_loader, valid_loader = get_distributed_loader(train_dataset, valid_dataset, world_size, rank, 0, batch_size)
    train_loader.sampler.set_epoch(epoch)
    model.train()
    agg = torch.zeros(4, device=device)
    for data, target in train_loader:
        data, target = data.to(device), target.to(device)
        loss, output, ok = dp_train_step(model, reducer, optimizer, loss_fn, data, target)
        use = (ok.reshape(()) == 1.0).to(torch.float32)      # skipped batches do not count (device-side, no sync)
        pred = output.argmax(1)
        agg[0] += torch.nan_to_num(_per_sample(loss, loss_fn, data.size(0))) * use
        agg[1] += (pred == target).float().mean() * use; agg[2] += use
    if scheduler:
        scheduler.step()
    model.eval()
    vag = torch.zeros(3, device=device)
    with torch.no_grad():
        for data, target in valid_loader:
            data, target = data.to(device), target.to(device)
            output = model(data)
            loss = loss_fn(output, target)
            vag[0] += _per_sample(loss, loss_fn, data.size(0)); vag[1] += (output.argmax(1) == target).float().mean(); vag[2] += 1
    dist.all_reduce(agg); dist.all_reduce(vag)
    n, vn = max(float(agg[2]), 1.0), max(float(vag[2]), 1.0)
    return float(agg[0]) / n, float(agg[1]) / n, float(vag[0]) / vn, float(vag[1]) / vn


def train_per_proc(rank: int, world_size: int, batch_size: Optional[int], model: torch.nn.Module, train_dataset: Dataset,
                   valid_dataset: Dataset, random_seed: int = 42, resume: bool = True, loss_fn=None,
                   model_filepath: str = "./weights/distributed.pt", num_epoch: int = 64, verbose: Optional[int] = 8,
                   save_best_only: bool = False, save_best_dir: str = "./weights/best.pt"):
    """Per-process body (reference :113-186): RCCL process group, broadcast, epochs, rank-0 checkpoint."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    use_gpu = torch.cuda.is_available()
    dist.init_process_group("nccl" if use_gpu else "gloo", rank=rank, world_size=world_size)
    device = torch.device(f"cuda:{rank}") if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    set_random_seeds(random_seed)
    model.to(device)
    if resume and os.path.isfile(model_filepath):
        model.load_state_dict(torch.load(model_filepath, map_location=device, weights_only=True), strict=False)
    broadcast_module_state(model, 0)
    reducer = GradAllReducer(model)
    if use_gpu:
        from .optim import ClipAdamW                         # AdamW(lr 2e-4) of reference :51, with the device-side skip
        optimizer = ClipAdamW(model.parameters(), lr=2e-4)
    else:
        optimizer = torch.optim.AdamW(model.parameters(), lr=2e-4)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(optimizer, T_0=8, T_mult=2)
    hist = {k: [] for k in ("train_loss", "train_acc", "valid_loss", "valid_acc")}
    best = float("inf")
    for epoch in range(num_epoch):
        tl, ta, vl, va = train_epoch_per_procs(rank, world_size, batch_size, model, train_dataset, valid_dataset, random_seed,
                                               resume, loss_fn, model_filepath, optimizer, scheduler, reducer, epoch)
        dist.barrier()
        if rank == 0:
            for k, v in zip(hist, (tl, ta, vl, va)):
                hist[k].append(v)
            if verbose and epoch % verbose == 0:
                print(f"epoch {epoch + 1}: train loss {tl:.3f} acc {ta:.3f} | valid loss {vl:.3f} acc {va:.3f}")
            if not save_best_only:               # the per-epoch "last" checkpoint (reference :166-178)
                os.makedirs(os.path.dirname(model_filepath) or ".", exist_ok=True)
                torch.save(model.state_dict(), model_filepath)
            if vl < best:
                best = vl
                os.makedirs(os.path.dirname(save_best_dir) or ".", exist_ok=True)
                torch.save(model.state_dict(), save_best_dir)
    dist.barrier()
    dist.destroy_process_group()
    return hist


def train_distributed(world_size: int, batch_size: Optional[int], model: torch.nn.Module, train_dataset: Dataset,
                      valid_dataset: Dataset, random_seed: int = 42, resume: bool = True, loss_fn=None,
                      model_filepath: str = "./weights/distributed.pt", num_epoch: int = 64, verbose: Optional[int] = 8,
                      save_best_only: bool = False, save_best_dir: str = "./weights/best.pt"):
    """Spawn one process per GPU (reference :189-213)."""
    # three busy streams per rank (compute chain, weight gradients, RCCL): with the HIP runtime's default of 4 hardware queues the
    # RCCL stream's cross-stream waits stall the compute chain at every stage boundary (bench.py: 6.54 vs 6.29 ms per step); the
    # children inherit the setting and read it when their runtime initialises
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    mp.spawn(train_per_proc,
             args=(world_size, batch_size, model, train_dataset, valid_dataset, random_seed, resume, loss_fn, model_filepath,
                   num_epoch, verbose, save_best_only, save_best_dir),
             nprocs=world_size, join=True)

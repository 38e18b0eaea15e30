"""Gradient clipping + AdamW for the whole model in three launches.

The reference's step is ``torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm_grad)`` followed by
``optimizer.step()`` with ``torch.optim.AdamW`` (src/train.py:64-66, train_vision_network.py:277-278).  On ~100
parameter tensors that is a chain of multi-tensor launches (~0.4 ms per step on MI355X); ``ClipAdamW`` does the same
arithmetic through ``md_opt_grad_norm`` + ``md_opt_adamw_step`` (include/mi355x_disrupt.h), which walk a device table of
all tensors.  Hyper-parameters, defaults and the update rule are those of ``torch.optim.AdamW`` (no amsgrad, no
maximize); ``state_dict`` holds ``exp_avg`` / ``exp_avg_sq`` / ``step`` per parameter (``step`` as a Python int; the group's
``step`` mirrors the largest for information).

The training loops of this package (src/train.py, src/distributed.py) recognise ``fused_clip`` and hand the clip
threshold to ``step(max_norm=...)`` instead of calling ``clip_grad_norm_`` themselves.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _native as N

_TENSOR_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8")])
_CHUNK_DT = np.dtype([("tensor", "<i4"), ("offset", "<i4")])


class ClipAdamW(torch.optim.Optimizer):
    fused_clip = True

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_norm: Optional[float] = None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("ClipAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, step=0))
        self.max_norm = max_norm
        self.last_grad_norm: Optional[torch.Tensor] = None      # device scalar of the most recent step (no host sync)
        self._cache = {}                                         # group index -> tables (see _tables)
        self._pending_ok = []                                    # (flag, event, params) of steps issued with ok= (see _settle_skips)

    # ------------------------------------------------------------------ tables
    _RING = 4      # pinned staging buffers per group: the host may run this many table refreshes ahead of the device

    def _tables(self, slot_key, ps):
        """Device tables for one set of parameters that share a step count: (tensor table, chunk table, #chunks, scratch).

        The chunk table depends on the parameter sizes only.  The tensor table holds raw pointers; gradient tensors are
        re-created by autograd every step and some change address, and ``load_state_dict`` replaces the moment tensors, so
        the table is refreshed whenever ANY of the four pointers of a parameter moved -- through pinned staging buffers and
        an asynchronous copy on the current stream, never a host synchronisation."""
        for p in ps:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("ClipAdamW: parameters must be contiguous CUDA float32 (no CPU fallback)")
            g = p.grad
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                raise RuntimeError("ClipAdamW: gradients must be contiguous CUDA float32 (no CPU fallback)")
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr())
                    for p in ps)
        hit = self._cache.get(slot_key)
        if hit is not None and hit["key"] == key:
            return hit
        ids = tuple(id(p) for p in ps)
        if hit is None or hit["ids"] != ids:
            dev = ps[0].device
            chunk = N.lib().md_opt_chunk_elems()
            tens = np.zeros(len(ps), dtype=_TENSOR_DT)
            chunks = []
            for i, p in enumerate(ps):
                tens[i] = (0, 0, 0, 0, p.numel())
                chunks.extend((i, o) for o in range((p.numel() + chunk - 1) // chunk))
            ch = np.array(chunks, dtype=_CHUNK_DT)
            nbytes = tens.nbytes
            hit = {
                "ids": ids, "host": tens, "nch": len(chunks),
                "tens": torch.empty(nbytes, dtype=torch.uint8, device=dev),
                "chunks": torch.from_numpy(ch.view(np.uint8).copy()).to(dev),
                "partial": torch.empty(len(chunks) + 2, device=dev, dtype=torch.float32),     # [norm, coef | per-chunk sums]
                "pinned": [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(self._RING)],
                "events": [None] * self._RING, "slot": 0,
            }
            self._cache[slot_key] = hit
        tens = hit["host"]
        tens["p"] = [k[0] for k in key]
        tens["g"] = [k[1] for k in key]
        tens["m"] = [k[2] for k in key]
        tens["v"] = [k[3] for k in key]
        slot = hit["slot"]; hit["slot"] = (slot + 1) % self._RING
        if hit["events"][slot] is not None:
            hit["events"][slot].synchronize()             # this staging buffer's previous copy (4 refreshes ago) is done
        pin = hit["pinned"][slot]
        pin.numpy()[:] = tens.view(np.uint8)
        hit["tens"].copy_(pin, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(); hit["events"][slot] = ev
        hit["key"] = key
        return hit

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._cache.clear()                                   # the moment tensors were replaced

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_cache"):
            self._cache.clear()

    # ------------------------------------------------------------------ step
    def _settle_skips(self, block: bool = False) -> None:
        """Steps issued with ``ok=`` advance the host-side step counters optimistically; once the device flag of such a step is
        known (its event has completed -- never waited for inside the training loop) and says "skipped", the counters are taken
        back.  Until then the bias correction of the next step is one count ahead: exceptional path only (non-finite loss)."""
        keep = []
        for flag, ev, params in self._pending_ok:
            if block:
                ev.synchronize()
            if ev.query():
                if float(flag[0]) != 1.0:                       # pinned host copy, complete: reading it touches no stream
                    for p in params:
                        self.state[p]["step"] = max(0, int(self.state[p]["step"]) - 1)
            else:
                keep.append((flag, ev, params))
        self._pending_ok = keep

    def state_dict(self):
        self._settle_skips(block=True)
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None, max_norm: Optional[float] = None, ok: Optional[torch.Tensor] = None):
        """``ok`` (optional device scalar, float32): the update is applied only if it equals 1 (see md_opt_adamw_step_if)."""
        if self._pending_ok:
            self._settle_skips()
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        max_norm = self.max_norm if max_norm is None else max_norm
        L = N.lib()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # Parameters that received a gradient, partitioned by their own step count (torch.optim.AdamW keeps ``step`` per
        # parameter: one that gets its first gradient late starts its bias correction at 1).  Normally one partition per group.
        work = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            by_step = {}
            for p in ps:
                by_step.setdefault(int(self.state[p].get("step", 0)), []).append(p)
            for st_count, sub in sorted(by_step.items()):
                work.append((gi, group, st_count, sub))
        groups_with_work = sorted({gi for gi, _, _, _ in work})
        if max_norm and len(groups_with_work) > 1:
            raise RuntimeError("ClipAdamW: gradient clipping across several parameter groups is not supported")
        coef = None
        if max_norm and work:
            # one norm over every parameter that has a gradient (all partitions of the group)
            allp = [p for _, _, _, sub in work for p in sub]
            tb = self._tables((work[0][0], -1), allp)
            partial = tb["partial"]
            N.check(L.md_opt_grad_norm(C.c_void_p(tb["tens"].data_ptr()), C.c_void_p(tb["chunks"].data_ptr()), tb["nch"],
                                       float(max_norm), C.c_void_p(partial[2:].data_ptr()), C.c_void_p(partial.data_ptr()), stream),
                    "md_opt_grad_norm")
            coef = C.c_void_p(partial.data_ptr())
            self.last_grad_norm = partial[0]
        for gi, group, st_count, sub in work:
            single = sum(1 for w in work if w[0] == gi) == 1
            tb = self._tables((gi, -1 if single else st_count), sub)
            for p in sub:
                self.state[p]["step"] = st_count + 1
            b1, b2 = group["betas"]
            N.check(L.md_opt_adamw_step_if(C.c_void_p(tb["tens"].data_ptr()), C.c_void_p(tb["chunks"].data_ptr()), tb["nch"], coef,
                                           float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           float(group["weight_decay"]), int(st_count + 1),
                                           None if ok is None else C.c_void_p(ok.data_ptr()), stream), "md_opt_adamw_step_if")
            group["step"] = max(int(group.get("step", 0)), st_count + 1)
        if ok is not None and work:
            # the flag is copied to pinned host memory behind this step (asynchronously) and inspected only once the event says
            # the copy has landed: a ``.item()`` on the device tensor would wait for everything queued on the stream by then --
            # the whole next forward and backward -- and serialise host and GPU (measured: 3.9 ms per step)
            host_flag = torch.empty(1, dtype=torch.float32, pin_memory=True)
            host_flag.copy_(ok.reshape(1).to(torch.float32), non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
            self._pending_ok.append((host_flag, ev, [p for _, _, _, sub in work for p in sub]))
        return loss

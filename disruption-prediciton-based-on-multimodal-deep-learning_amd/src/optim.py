"""Gradient clipping + AdamW for the whole model in three launches.

The reference's step is ``torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm_grad)`` followed by
``optimizer.step()`` with ``torch.optim.AdamW`` (src/train.py:64-66, train_vision_network.py:277-278).  On ~100
parameter tensors that is a chain of multi-tensor launches (~0.4 ms per step on MI355X); ``ClipAdamW`` does the same
arithmetic through ``md_opt_grad_norm`` + ``md_opt_adamw_step`` (include/mi355x_disrupt.h), which walk a device table of
all tensors.  Hyper-parameters, defaults and the update rule are those of ``torch.optim.AdamW`` (no amsgrad, no
maximize); ``state_dict`` holds ``exp_avg`` / ``exp_avg_sq`` per parameter and ``step`` per group.

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

    # ------------------------------------------------------------------ tables
    _RING = 4      # pinned staging buffers per group: the host may run this many table refreshes ahead of the device

    def _tables(self, gi: int, group):
        """Device tables for one parameter group: (key, tensor table, chunk table, #chunks, scratch).

        The chunk table depends on the parameter sizes only.  The tensor table holds raw pointers; gradient tensors are
        re-created by autograd every step and some change address, so the table is refreshed whenever a pointer moved --
        through pinned staging buffers and an asynchronous copy on the current stream, never a host synchronisation."""
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return None
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        hit = self._cache.get(gi)
        if hit is not None and hit["key"] == key:
            return hit
        ids = tuple(id(p) for p in ps)
        if hit is None or hit["ids"] != ids:
            dev = ps[0].device
            chunk = N.lib().md_opt_chunk_elems()
            tens = np.zeros(len(ps), dtype=_TENSOR_DT)
            chunks = []
            for i, p in enumerate(ps):
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("ClipAdamW: parameters must be contiguous CUDA float32 (no CPU fallback)")
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                tens[i] = (0, 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
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
            self._cache[gi] = hit
        for p in ps:
            g = p.grad
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                raise RuntimeError("ClipAdamW: gradients must be contiguous CUDA float32 (no CPU fallback)")
        tens = hit["host"]
        tens["p"] = [k[0] for k in key]
        tens["g"] = [k[1] for k in key]
        slot = hit["slot"]; hit["slot"] = (slot + 1) % self._RING
        if hit["events"][slot] is not None:
            hit["events"][slot].synchronize()             # this staging buffer's previous copy (4 refreshes ago) is done
        pin = hit["pinned"][slot]
        pin.numpy()[:] = tens.view(np.uint8)
        hit["tens"].copy_(pin, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(); hit["events"][slot] = ev
        hit["key"] = key
        return hit

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None, max_norm: Optional[float] = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        max_norm = self.max_norm if max_norm is None else max_norm
        L = N.lib()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        live = [(g, self._tables(i, g)) for i, g in enumerate(self.param_groups)]
        live = [(g, t) for g, t in live if t is not None]
        if max_norm and len(live) > 1:
            raise RuntimeError("ClipAdamW: gradient clipping across several parameter groups is not supported")
        for group, tb in live:
            tens, chunks, nch, partial = tb["tens"], tb["chunks"], tb["nch"], tb["partial"]
            coef = None
            if max_norm:
                N.check(L.md_opt_grad_norm(C.c_void_p(tens.data_ptr()), C.c_void_p(chunks.data_ptr()), nch, float(max_norm),
                                           C.c_void_p(partial[2:].data_ptr()), C.c_void_p(partial.data_ptr()), stream),
                        "md_opt_grad_norm")
                coef = C.c_void_p(partial.data_ptr())
                self.last_grad_norm = partial[0]
            group["step"] += 1
            b1, b2 = group["betas"]
            N.check(L.md_opt_adamw_step(C.c_void_p(tens.data_ptr()), C.c_void_p(chunks.data_ptr()), nch, coef,
                                        float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                        float(group["weight_decay"]), int(group["step"]), stream), "md_opt_adamw_step")
        return loss

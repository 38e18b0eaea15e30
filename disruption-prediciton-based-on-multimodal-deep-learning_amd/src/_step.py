"""One training step of ``R2Plus1DClassifier`` as ONE call into the C ABI (``md_plan_train_step``, csrc/step.hip).

The reference's step (src/train.py:40-66) is ``optimizer.zero_grad -> model(data) -> loss_fn(output, target) -> isfinite check ->
loss.backward -> clip_grad_norm_ -> optimizer.step``: on the MI355X path that is ~340 kernel launches which the composed form
issues through four ``torch.autograd.Function`` hops, three ``ctypes`` calls into the executor and the optimizer -- 2.3 ms of host
time per step (median; 9 ms at the 90th percentile, profiles/r03x_bench.json).  ``FusedTrainStep`` hands the whole step to
C: same kernels, same order, same stream, bit-identical parameters (tests/test_fused_step_gpu.py), no autograd engine.

What stays visible to the caller, as after the composed step:
  * ``p.grad`` of every parameter (views of two flat buffers owned by this object, rewritten by every step);
  * the optimizer's state (``exp_avg`` / ``exp_avg_sq`` / ``step``) -- ``ClipAdamW``'s own tables are used;
  * BatchNorm running statistics and ``num_batches_tracked``;
  * ``loss_fn.last_pred``.
The reference's finite-loss guard is the device flag ``ok`` (1.0 = applied): the update is gated on the device, the host reads the
flags whenever it wants to (``train_per_epoch``: once per epoch).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _native as N
from .loss import CELoss, FocalLoss, LDAMLoss, _dev_weight
from .ops import KIND


def applicable(model, loss_fn, optimizer) -> bool:
    """The fused step covers: R2Plus1DClassifier in training mode with the fused head and no module / tensor hooks, one of the three
    softmax losses, and a ClipAdamW with ONE parameter group that holds exactly the model's parameters (all requiring a gradient)."""
    from .models.R2Plus1D import R2Plus1DClassifier
    from .optim import ClipAdamW
    if type(model) is not R2Plus1DClassifier or type(optimizer) is not ClipAdamW or type(loss_fn) not in (FocalLoss, LDAMLoss, CELoss):
        return False
    if model.res2plus1d.grad_segment_hook is not None or len(optimizer.param_groups) != 1 or not model.training:
        return False
    # module hooks observe forward / backward calls that the fused step does not make: with any hook installed the composed step runs
    import torch.nn.modules.module as _mm
    if _mm._global_forward_hooks or _mm._global_forward_pre_hooks or _mm._global_backward_hooks or _mm._global_backward_pre_hooks:
        return False
    for mod in model.modules():
        if mod._forward_hooks or mod._forward_pre_hooks or mod._backward_hooks or mod._backward_pre_hooks:
            return False
    if any(getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None) for p in model.parameters()):
        return False
    ps = list(model.parameters())
    if any(not p.requires_grad or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() for p in ps):
        return False
    if {id(p) for p in ps} != {id(p) for p in optimizer.param_groups[0]["params"]} or len(ps) != len(optimizer.param_groups[0]["params"]):
        return False
    lin0 = model.linear[0]
    return lin0.in_features * lin0.out_features <= 65536


class FusedTrainStep:
    def __init__(self, model, loss_fn, optimizer):
        if not applicable(model, loss_fn, optimizer):
            raise RuntimeError("FusedTrainStep: this (model, loss, optimizer) triple is not covered; use the composed step")
        self.model, self.loss_fn, self.opt = model, loss_fn, optimizer
        self._units = model.res2plus1d.unit_modules()
        self._per_shape = {}
        dev = next(model.parameters()).device
        n = len(self._units)
        lin0, bn, _, lin1 = model.linear[0], model.linear[1], model.linear[2], model.linear[3]
        # flat gradient buffers: trunk (units in order: every w, every gamma, every beta -- the layout TrunkFunction.backward hands
        # to the data-parallel bucket) and head
        self._tparams = [u.conv.weight for u in self._units] + [u.bn.weight for u in self._units] + [u.bn.bias for u in self._units]
        self._hparams = [lin0.weight, lin0.bias, bn.weight, bn.bias, lin1.weight, lin1.bias]
        self._flat_t = torch.zeros(sum(p.numel() for p in self._tparams), device=dev, dtype=torch.float32)
        self._flat_h = torch.zeros(sum(p.numel() for p in self._hparams), device=dev, dtype=torch.float32)
        self._tgrads, o = [], 0
        for p in self._tparams:
            self._tgrads.append(self._flat_t[o:o + p.numel()].view(p.shape)); o += p.numel()
        self._hgrads, o = [], 0
        for p in self._hparams:
            self._hgrads.append(self._flat_h[o:o + p.numel()].view(p.shape)); o += p.numel()
        counters = [u.bn.num_batches_tracked for u in self._units] + [bn.num_batches_tracked]
        self._counters = counters
        self._counter_ptrs = [c.data_ptr() for c in counters]
        self._counter_tab = torch.tensor(self._counter_ptrs, dtype=torch.int64).to(dev)
        self._n = n

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _ptrs(tensors):
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def _bind_grads(self):
        for p, g in zip(self._tparams, self._tgrads):
            if p.grad is not g:
                p.grad = g
        for p, g in zip(self._hparams, self._hgrads):
            if p.grad is not g:
                p.grad = g

    def _buffers(self, x):
        key = tuple(x.shape)
        hit = self._per_shape.get(key)
        if hit is None:
            B, _, T, H, W = x.shape
            plan = self.model.res2plus1d._plan(B, T, H, W)
            lin0, lin1 = self.model.linear[0], self.model.linear[3]
            D, Hd, K = lin0.in_features, lin0.out_features, lin1.out_features
            dev = x.device
            f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)      # noqa: E731
            hit = dict(plan=plan, ws=plan.new_workspace(dev), feat=f(B, D), dfeat=f(B, D), logits=f(B, K), dlogits=f(B, K),
                       save=f(N.lib().md_head_save_floats(B, D, Hd)), D=D, Hd=Hd, K=K)
            self._per_shape[key] = hit
        return hit

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def __call__(self, x: torch.Tensor, target: torch.Tensor, max_norm: Optional[float] = None, update: bool = True):
        """Returns (loss 0-d, logits (B,K), pred (B,) int64, ok 0-d float) -- fresh tensors except ``logits`` (per-shape buffer,
        overwritten by the next step of that shape)."""
        m, opt = self.model, self.opt
        if not m.training:
            raise RuntimeError("FusedTrainStep: the model is in eval mode")
        if x.dim() != 5 or x.size(1) != 3 or not x.is_cuda:
            raise RuntimeError(f"FusedTrainStep expects a CUDA (B,3,T,H,W) clip, got {tuple(x.shape)} on {x.device}")
        x = x.contiguous().float()
        target = target.contiguous().view(-1)
        if target.dtype != torch.int64 or target.numel() != x.size(0):
            raise RuntimeError("target must be int64 of length B")
        from . import ops
        ops.check_fp16_range(x, "the clip")
        hb = self._buffers(x)
        self._bind_grads()
        u = self._units
        lin0, bn, elu, lin1 = m.linear[0], m.linear[1], m.linear[2], m.linear[3]
        lf = self.loss_fn
        dev = x.device
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        pred = torch.empty(x.size(0), device=dev, dtype=torch.int64)
        ok = torch.empty(1, device=dev, dtype=torch.float32)
        a = N.MdTrainStepArgs()
        a.B, a.Hd, a.K = x.size(0), hb["Hd"], hb["K"]
        if type(lf) is FocalLoss:
            a.loss_kind, a.gamma_or_s, margins = KIND["focal"], float(lf.gamma), None
        elif type(lf) is LDAMLoss:
            a.loss_kind, a.gamma_or_s, margins = KIND["ldam"], float(lf.s), _dev_weight(lf.m_list, dev)
        else:
            a.loss_kind, a.gamma_or_s, margins = KIND["ce"], 0.0, None
        cw = _dev_weight(lf.weight, dev)
        n = self._n
        keep = [self._ptrs(self._tparams[:n]), self._ptrs(self._tparams[n:2 * n]), self._ptrs(self._tparams[2 * n:]),
                self._ptrs([v.bn.running_mean for v in u]), self._ptrs([v.bn.running_var for v in u]),
                self._ptrs(self._tgrads[:n]), self._ptrs(self._tgrads[n:2 * n]), self._ptrs(self._tgrads[2 * n:])]
        vp = lambda arr: C.cast(arr, C.c_void_p)      # noqa: E731
        a.x, a.target = x.data_ptr(), target.data_ptr()
        a.w, a.gamma, a.beta, a.rmean, a.rvar, a.dw, a.dgamma, a.dbeta = [vp(k) for k in keep]
        a.w0, a.b0, a.hgamma, a.hbeta, a.w1, a.b1 = [p.data_ptr() for p in self._hparams]
        a.hrmean, a.hrvar = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        a.dw0, a.db0, a.dhgamma, a.dhbeta, a.dw1, a.db1 = [g.data_ptr() for g in self._hgrads]
        a.head_alpha, a.head_eps, a.head_momentum = float(elu.alpha), float(bn.eps), float(bn.momentum)
        a.class_weight = None if cw is None else cw.data_ptr()
        a.margins = None if margins is None else margins.data_ptr()
        a.feat, a.dfeat, a.logits, a.dlogits = hb["feat"].data_ptr(), hb["dfeat"].data_ptr(), hb["logits"].data_ptr(), hb["dlogits"].data_ptr()
        a.head_save, a.loss, a.pred, a.workspace = hb["save"].data_ptr(), loss.data_ptr(), pred.data_ptr(), hb["ws"].data_ptr()
        ptrs = [c.data_ptr() for c in self._counters]      # (load_state_dict copies in place; a replaced buffer would move)
        if ptrs != self._counter_ptrs:
            self._counter_ptrs = ptrs
            self._counter_tab = torch.tensor(ptrs, dtype=torch.int64).to(dev)
        a.counters, a.ncounters = self._counter_tab.data_ptr(), len(self._counters)
        a.ok_flag = ok.data_ptr()
        params = None
        if update:
            if opt._pending_ok:
                opt._settle_skips()
            group = opt.param_groups[0]
            params = list(group["params"])
            steps = {int(opt.state[p].get("step", 0)) for p in params}
            if len(steps) != 1:
                raise RuntimeError("FusedTrainStep: parameters with different step counts (use optimizer.step for this model)")
            st = steps.pop()
            tb = opt._tables((0, -1), params)
            b1, b2 = group["betas"]
            mn = opt.max_norm if max_norm is None else max_norm
            a.opt_nchunks = tb["nch"]
            a.opt_tensors, a.opt_chunks, a.opt_partial = tb["tens"].data_ptr(), tb["chunks"].data_ptr(), tb["partial"].data_ptr()
            a.max_norm = float(mn) if mn else 0.0
            a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"])
            a.opt_step = st + 1
            for p in params:
                opt.state[p]["step"] = st + 1
            group["step"] = max(int(group.get("step", 0)), st + 1)
            if mn:
                opt.last_grad_norm = tb["partial"][0]
        else:
            a.opt_nchunks = 0
        N.check(N.lib().md_plan_train_step(hb["plan"]._h, C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                "md_plan_train_step")
        if update:
            # as ClipAdamW.step(ok=...): the flag lands in pinned host memory behind the step; the optimizer takes the step count
            # back once it knows the update was skipped
            host_flag = torch.empty(1, dtype=torch.float32, pin_memory=True)
            host_flag.copy_(ok, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
            opt._pending_ok.append((host_flag, ev, params))
        lf.last_pred = pred
        return loss.view(()), hb["logits"], pred, ok.view(())

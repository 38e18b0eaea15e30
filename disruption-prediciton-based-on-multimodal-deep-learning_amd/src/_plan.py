"""Python handle of the C++ trunk executor (MdPlan, include/mi355x_disrupt.h) + its autograd bridge.

One plan per (B,T,H,W); the plan is host metadata only, the activation workspace is a torch uint8
tensor (caching allocator) that lives as long as the autograd node that needs it for backward.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from . import _native as N
from .ops import _stream, require_cuda


def _ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class TrunkPlan:
    def __init__(self, B: int, T: int, H: int, W: int, layer_sizes: Sequence[int], alpha: float):
        L = N.lib()
        self.shape = (B, T, H, W)
        self.alpha = float(alpha)
        ls = (C.c_int32 * 4)(*[int(v) for v in layer_sizes])
        h = C.c_void_p()
        N.check(L.md_plan_create(B, T, H, W, ls, float(alpha), C.byref(h)), "md_plan_create")
        self._h = h
        self.num_units = L.md_plan_num_units(h)
        self.descs: List[N.MdConvDesc] = []
        for i in range(self.num_units):
            d = N.MdConvDesc()
            N.check(L.md_plan_unit_desc(h, i, C.byref(d)), "md_plan_unit_desc")
            self.descs.append(d)
        self.workspace_bytes = L.md_plan_workspace_bytes(h)
        self.feat_dim = L.md_plan_feat_dim(h)
        self._eval_ws = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                N.lib().md_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def use_side_stream(self, on: bool) -> None:
        """Backward schedule (md_plan_use_side_stream): weight gradients on a side stream, or everything on one stream (default)."""
        N.check(N.lib().md_plan_use_side_stream(self._h, int(bool(on))), "md_plan_use_side_stream")

    def side_stream(self):
        """The plan's side stream as a torch stream (None when there is none): see md_plan_defer_join."""
        h = N.lib().md_plan_side_stream(self._h)
        return torch.cuda.ExternalStream(h) if h else None

    def defer_join(self, on: bool) -> None:
        N.check(N.lib().md_plan_defer_join(self._h, int(bool(on))), "md_plan_defer_join")

    def join(self) -> None:
        N.check(N.lib().md_plan_join(self._h, _stream()), "md_plan_join")

    def profile_enable(self, on, keep: bool = False) -> None:
        """HIP events around every conv launch; keep=True resumes without forgetting earlier records (step sampling)."""
        N.check(N.lib().md_plan_profile_enable(self._h, (2 if keep else 1) if on else 0), "md_plan_profile_enable")

    def profile_reserve(self, records: int) -> None:
        """Create the event pairs for that many bracketed launches ahead of a timed region."""
        N.check(N.lib().md_plan_profile_reserve(self._h, int(records)), "md_plan_profile_reserve")

    def profile_read(self):
        """[(ms, launches, flops)] for conv forward / data-gradient / weight-gradient since the last read."""
        ms = (C.c_double * 3)(); ln = (C.c_int64 * 3)(); fl = (C.c_double * 3)()
        N.check(N.lib().md_plan_profile_read(self._h, ms, ln, fl), "md_plan_profile_read")
        return [(ms[i], ln[i], fl[i]) for i in range(3)]

    def weight_shape(self, i: int):
        d = self.descs[i]
        return (d.Cout, d.Cin, d.kt, d.kh, d.kw)

    def unit_tensors(self, ws: torch.Tensor, i: int):
        """Diagnostics: views into a workspace a forward has filled -- (raw conv output [rows, Cp], stats [4, Cp] =
        mean | invstd | scale | shift) of unit i."""
        ro, so, rows, cp = C.c_size_t(), C.c_size_t(), C.c_int64(), C.c_int32()
        N.check(N.lib().md_plan_unit_layout(self._h, i, C.byref(ro), C.byref(so), C.byref(rows), C.byref(cp)), "md_plan_unit_layout")
        f = ws.view(torch.float32)
        raw = f[ro.value:ro.value + rows.value * cp.value].view(rows.value, cp.value)
        st = f[so.value:so.value + 4 * cp.value].view(4, cp.value)
        return raw, st

    def z_tensor(self, ws: torch.Tensor, zi: int):
        """Diagnostics: materialised tensor zi of a filled workspace, [rows, cpad(C)], and C."""
        off, rows, cc = C.c_size_t(), C.c_int64(), C.c_int32()
        N.check(N.lib().md_plan_z_layout(self._h, zi, C.byref(off), C.byref(rows), C.byref(cc)), "md_plan_z_layout")
        cp = (cc.value + 3) & ~3
        return ws.view(torch.float32)[off.value:off.value + rows.value * cp].view(rows.value, cp), cc.value

    def new_workspace(self, device) -> torch.Tensor:
        return torch.empty(self.workspace_bytes, dtype=torch.uint8, device=device)

    def eval_workspace(self, device) -> torch.Tensor:
        if self._eval_ws is None or self._eval_ws.device != device:
            self._eval_ws = self.new_workspace(device)
        return self._eval_ws

    def forward(self, x, ws, weights, gammas, betas, rmeans, rvars, training: bool) -> torch.Tensor:
        B = self.shape[0]
        feat = torch.empty((B, self.feat_dim), device=x.device, dtype=torch.float32)
        N.check(N.lib().md_plan_forward(self._h, x.data_ptr(), _ptr_array(weights), _ptr_array(gammas), _ptr_array(betas),
                                        _ptr_array(rmeans), _ptr_array(rvars), int(training), feat.data_ptr(),
                                        ws.data_ptr(), _stream()), "md_plan_forward")
        return feat

    def backward_range(self, dfeat, ws, weights, gammas, dws, dgammas, dbetas, hi: int = 4, lo: int = 0) -> None:
        N.check(N.lib().md_plan_backward_range(self._h, None if dfeat is None else dfeat.data_ptr(), _ptr_array(weights),
                                               _ptr_array(gammas), _ptr_array(dws), _ptr_array(dgammas),
                                               _ptr_array(dbetas), ws.data_ptr(), hi, lo, _stream()),
                "md_plan_backward_range")


class TrunkFunction(torch.autograd.Function):
    """feat = R2Plus1DNet(x); parameters are passed flat as [w_0.., gamma_0.., beta_0..]."""

    @staticmethod
    def forward(ctx, plan: TrunkPlan, x, rmeans, rvars, training, need_bwd, seg_hook, *params):
        """``need_bwd`` is decided by the CALLER (grad mode is always off inside an autograd.Function's forward): when a
        backward may follow, the activations live in a workspace of their own that this node owns until its backward has
        run; otherwise the plan's cached scratch workspace is used."""
        n = plan.num_units
        weights, gammas, betas = params[:n], params[n:2 * n], params[2 * n:3 * n]
        require_cuda(x, *params)
        ws = plan.new_workspace(x.device) if need_bwd else plan.eval_workspace(x.device)
        feat = plan.forward(x, ws, weights, gammas, betas, rmeans, rvars, training)
        ctx.plan = plan
        ctx.ws = ws if need_bwd else None
        ctx.training = bool(training)
        ctx.seg_hook = seg_hook
        ctx.params = params
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        if not ctx.training:
            raise RuntimeError("R2Plus1DNet: backward through an eval-mode forward is not supported on the MI355X path "
                               "(BatchNorm uses running statistics there; call model.train() or wrap the forward in no_grad)")
        if ctx.ws is None:
            raise RuntimeError("R2Plus1DNet: this forward was run without saving activations (no parameter required a gradient)")
        plan: TrunkPlan = ctx.plan
        n = plan.num_units
        params = ctx.params
        weights, gammas = params[:n], params[n:2 * n]
        dev = dfeat.device
        dfeat = dfeat.contiguous()
        # one flat gradient buffer (units in order: w, gamma, beta): a single bucket for the DP all-reduce
        sizes = [p.numel() for p in params]
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        grads, o = [], 0
        for p, s in zip(params, sizes):
            grads.append(flat[o:o + s].view(p.shape)); o += s
        dws, dgs, dbs = grads[:n], grads[n:2 * n], grads[2 * n:3 * n]
        if ctx.seg_hook is None:
            plan.backward_range(dfeat, ctx.ws, weights, gammas, dws, dgs, dbs, 4, 0)
        else:
            # stage-wise gradient exchange: a stage's weight gradients are produced on the plan's side stream; the hook
            # queues their all-reduce behind THAT stream, so the collective waits for them and the backward chain on the
            # current stream does not.  One join at the end, before the BatchNorm gradients / the handles are consumed.
            side = plan.side_stream()
            plan.defer_join(side is not None)
            try:
                for st in (4, 3, 2, 1, 0):
                    plan.backward_range(dfeat, ctx.ws, weights, gammas, dws, dgs, dbs, st, st)
                    if st == 0:
                        plan.join()
                    ctx.seg_hook(st, flat, grads, side if st > 0 else None)
            finally:
                plan.defer_join(False)
        ctx.ws = None
        return (None, None, None, None, None, None, None) + tuple(grads)

"""Composable per-unit autograd bridges over the C ABI: one Conv3d+BatchNorm3d+LeakyReLU unit
(reference Conv3dBlock, src/models/R2Plus1D.py:25-58) and the classifier head (:243-248).

The R(2+1)D trunk does not use these (it runs as one executor plan, ``_plan.py``); they exist so that
sub-modules called on their own -- and other encoders assembled from the same unit -- run on the same
gfx950 kernels.  Tensors cross this boundary in the reference's (B,C,T,H,W) layout.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _native as N
from .. import ops
from ..utils import streams


# BatchNorm's num_batches_tracked is a device tensor: ``+= 1`` per unit is one tiny launch per BatchNorm per step (50 in SlowFast).
# Inside ``deferred_bn_counters()`` -- the models' forward() -- the increments are collected and applied by ONE _foreach_add_.
_counter_depth = 0
_counter_pending = []


class deferred_bn_counters:
    def __enter__(self):
        global _counter_depth
        _counter_depth += 1
        return self

    def __exit__(self, *exc):
        global _counter_depth
        _counter_depth -= 1
        if _counter_depth == 0 and _counter_pending:
            pend = list(_counter_pending)
            _counter_pending.clear()
            torch._foreach_add_(pend, 1)
        return False


class immediate_bn_counters:
    """Inside this context the counters are bumped at once even under an enclosing ``deferred_bn_counters()`` -- used while a
    branch is captured into a HIP graph (utils/graphed.py::GraphedBranch): the increment has to be a launch inside the graph."""

    def __enter__(self):
        global _counter_depth
        self.saved = _counter_depth
        _counter_depth = 0
        return self

    def __exit__(self, *exc):
        global _counter_depth
        _counter_depth = self.saved
        return False


def bump_batches_tracked(bn) -> None:
    if _counter_depth > 0:
        _counter_pending.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


class ConvBnLeakyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, gamma, beta, rmean, rvar, stride, padding, slope, training, eps, momentum, cl_channels=0):
        # cl_channels > 0: x is already the kernels' channels-last tensor (B,T,H,W,Cp) with that many channels, and the result
        # is handed on in the same layout (CLAct, below): no layout conversion at either end
        x = x.contiguous()                   # (strided views such as x[:, :, ::tau] are packed here: memory plumbing)
        ops.require_cuda(x, w, gamma, beta)
        ops.check_fp16_range(x, "a convolution's input")
        if cl_channels:
            B, T, H, W, _ = x.shape
            Cin = int(cl_channels)
            xcl = x.float()
        else:
            B, Cin, T, H, W = x.shape
            xcl = None
        Cout = w.shape[0]
        d = ops.make_desc(B, T, H, W, Cin, Cout, tuple(w.shape[2:]), tuple(stride), tuple(padding))
        if xcl is None:
            xcl = ops.to_channels_last(x.contiguous().float())
        wf, wd = ops.pack_weights(d, w.contiguous(), want_dgrad=training)
        y, part = ops.conv_fwd(d, ops.view(xcl), wf, x.device, want_stats=training)
        rows = y.numel() // y.shape[-1]
        if training:
            st = ops.bn_finalize(part, Cout, rows, gamma, beta, rmean, rvar, eps, momentum)
        else:
            st = torch.empty((4, ops.cpad(Cout)), device=x.device, dtype=torch.float32)
            N.check(N.lib().md_bn_eval_params(Cout, ops._p(gamma), ops._p(beta), ops._p(rmean), ops._p(rvar), eps,
                                              ops._p(st[0]), ops._p(st[1]), ops._p(st[2]), ops._p(st[3]), ops._stream()),
                    "md_bn_eval_params")
        a = ops.bn_act(ops.view(y, st[2], st[3], slope), y, Cout)
        out = a if cl_channels else ops.from_channels_last(a, Cout)
        if training:
            ctx.d = d
            ctx.slope = slope
            ctx.save_for_backward(xcl, y, st, wd)
        ctx.training = training
        ctx.cl = bool(cl_channels)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise RuntimeError("mi355x hot path: backward through an eval-mode BatchNorm unit is not supported")
        xcl, y, st, wd = ctx.saved_tensors
        d = ctx.d
        dA = dout.contiguous().float() if ctx.cl else ops.to_channels_last(dout.contiguous().float())
        d_raw, _, dgamma, dbeta = ops.bn_backward(dA, ops.view(y, st[2], st[3], ctx.slope), st, d.Cout)
        dx = None
        if ctx.needs_input_grad[0] and streams.unit_helpers(d_raw):
            with streams.fork(d_raw.device, None, (d_raw, xcl)) as f:      # weight gradient beside the data gradient
                dw = ops.conv_wgrad(d, ops.view(xcl), d_raw)
            dx = ops.conv_dgrad(d, d_raw, wd)
            f.join(dw)
        else:
            dw = ops.conv_wgrad(d, ops.view(xcl), d_raw)
            if ctx.needs_input_grad[0]:
                dx = ops.conv_dgrad(d, d_raw, wd)
        if dx is not None and not ctx.cl:
            dx = ops.from_channels_last(dx, d.Cin)
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None, None, None, None


class CLAct:
    """An activation kept in the kernels' channels-last layout between units: ``t`` (B,T,H,W,Cp) fp32 with the padding channels
    exactly zero, ``C`` real channels.  The native ResNet3D stages hand these from unit to unit (conv+BN units, Swish, the
    residual close are layout-agnostic or channels-last native), so the (B,C,T,H,W) <-> channels-last conversion kernels run
    at the stage boundaries that need the reference layout only (stem pooling, squeeze-excitation, the final pool)."""
    __slots__ = ("t", "C")

    def __init__(self, t: torch.Tensor, C: int):
        self.t, self.C = t, int(C)

    @property
    def shape(self):                         # the logical (B,C,T,H,W) shape
        B, T, H, W, _ = self.t.shape
        return (B, self.C, T, H, W)


class _ToCLFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.C = x.shape[1]
        return ops.to_channels_last(x.contiguous().float())

    @staticmethod
    def backward(ctx, dout):
        return ops.from_channels_last(dout.contiguous().float(), ctx.C)


class _FromCLFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xcl, C):
        return ops.from_channels_last(xcl.contiguous().float(), int(C))

    @staticmethod
    def backward(ctx, dout):
        return ops.to_channels_last(dout.contiguous().float()), None


def to_cl_act(x: torch.Tensor) -> CLAct:
    return CLAct(_ToCLFunction.apply(x), x.shape[1])


def from_cl_act(a: CLAct) -> torch.Tensor:
    return _FromCLFunction.apply(a.t, a.C)


class _CatCLFunction(torch.autograd.Function):
    """torch.cat on the channel axis of two channels-last activations, one launch each way (md_cat_cl / md_split_cl)."""

    @staticmethod
    def forward(ctx, a, Ca, b, Cb):
        a = ops.f32(a).contiguous(); b = ops.f32(b).contiguous()
        ops.require_cuda(a, b)
        assert a.shape[:-1] == b.shape[:-1] and a.shape[-1] == ops.cpad(Ca) and b.shape[-1] == ops.cpad(Cb)
        rows = a.numel() // a.shape[-1]
        out = torch.empty(a.shape[:-1] + (ops.cpad(Ca + Cb),), device=a.device, dtype=torch.float32)
        N.check(N.lib().md_cat_cl(ops._p(a), int(Ca), ops._p(b), int(Cb), rows, ops._p(out), ops._stream()), "md_cat_cl")
        ctx.Ca, ctx.Cb, ctx.sa, ctx.sb = int(Ca), int(Cb), a.shape, b.shape
        return out

    @staticmethod
    def backward(ctx, g):
        g = ops.f32(g).contiguous()
        da = torch.empty(ctx.sa, device=g.device, dtype=torch.float32)
        db = torch.empty(ctx.sb, device=g.device, dtype=torch.float32)
        rows = da.numel() // da.shape[-1]
        N.check(N.lib().md_split_cl(ops._p(g), ctx.Ca, ctx.Cb, rows, ops._p(da), ops._p(db), ops._stream()), "md_split_cl")
        return da, None, db, None


def cat_cl(a: CLAct, b: CLAct) -> CLAct:
    """torch.cat([a, b], dim=1) of the logical tensors, in the channels-last layout (padding channels zero)."""
    return CLAct(_CatCLFunction.apply(a.t, a.C, b.t, b.C), a.C + b.C)


def conv_bn_leaky(x, conv: torch.nn.Conv3d, bn: torch.nn.BatchNorm3d, slope: float, training: bool):
    if conv.bias is not None:
        raise NotImplementedError("mi355x hot path: Conv3dBlock with bias=True is not used by the reference")
    if isinstance(x, CLAct):
        out = ConvBnLeakyFunction.apply(x.t, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.stride,
                                        conv.padding, float(slope), bool(training), float(bn.eps), float(bn.momentum), x.C)
        if training:
            bump_batches_tracked(bn)
        return CLAct(out, conv.weight.shape[0])
    out = ConvBnLeakyFunction.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, conv.stride,
                                    conv.padding, float(slope), bool(training), float(bn.eps), float(bn.momentum))
    if training:
        bump_batches_tracked(bn)
    return out


class HeadFunction(torch.autograd.Function):
    """Linear(D->Hd) -> BatchNorm1d -> ELU(alpha) -> Linear(Hd->K) in one launch each way."""

    @staticmethod
    def forward(ctx, f, w0, b0, gamma, beta, w1, b1, rmean, rvar, alpha, eps, momentum, training):
        ops.require_cuda(f, w0, b0, gamma, beta, w1, b1)
        f = f.contiguous()
        B, D = f.shape
        Hd, K = w0.shape[0], w1.shape[0]
        L = N.lib()
        logits = torch.empty((B, K), device=f.device, dtype=torch.float32)
        save = torch.empty(L.md_head_save_floats(B, D, Hd), device=f.device, dtype=torch.float32)
        N.check(L.md_head_fwd(ops._p(f), B, D, Hd, K, ops._p(w0), ops._p(b0), ops._p(gamma), ops._p(beta), ops._p(w1),
                              ops._p(b1), alpha, eps, momentum, int(training), ops._p(rmean), ops._p(rvar), ops._p(logits),
                              ops._p(save), ops._stream()), "md_head_fwd")
        ctx.save_for_backward(f, w0, gamma, w1, save)
        ctx.alpha = alpha
        ctx.training = training
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        if not ctx.training:
            raise RuntimeError("mi355x hot path: backward through the eval-mode head is not supported")
        f, w0, gamma, w1, save = ctx.saved_tensors
        B, D = f.shape
        Hd, K = w0.shape[0], w1.shape[0]
        dev = f.device
        dlogits = dlogits.contiguous()
        df = torch.empty_like(f)
        dw0 = torch.empty_like(w0); db0 = torch.empty(Hd, device=dev)
        dg = torch.empty(Hd, device=dev); dbt = torch.empty(Hd, device=dev)
        dw1 = torch.empty_like(w1); db1 = torch.empty(K, device=dev)
        N.check(N.lib().md_head_bwd(ops._p(dlogits), ops._p(f), B, D, Hd, K, ops._p(w0), ops._p(gamma), ops._p(w1), ctx.alpha,
                                    ops._p(save), ops._p(df), ops._p(dw0), ops._p(db0), ops._p(dg), ops._p(dbt), ops._p(dw1),
                                    ops._p(db1), ops._stream()), "md_head_bwd")
        return df, dw0, db0, dg, dbt, dw1, db1, None, None, None, None, None, None


class ConvFunction(torch.autograd.Function):
    """Plain Conv3d (no bias, no normalisation): the SlowFast laterals (reference slowfast.py:58-65)."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, cl_channels=0):
        x = x.contiguous()
        ops.require_cuda(x, w)
        if cl_channels:
            B, T, H, W, _ = x.shape
            Cin = int(cl_channels)
            xcl = x.float()
        else:
            B, Cin, T, H, W = x.shape
            xcl = ops.to_channels_last(x.contiguous().float())
        Cout = w.shape[0]
        d = ops.make_desc(B, T, H, W, Cin, Cout, tuple(w.shape[2:]), tuple(stride), tuple(padding))
        wf, wd = ops.pack_weights(d, w.contiguous(), want_dgrad=True)
        y, _ = ops.conv_fwd(d, ops.view(xcl), wf, x.device, want_stats=False)
        ctx.d = d
        ctx.cl = bool(cl_channels)
        ctx.save_for_backward(xcl, wd)
        return y if cl_channels else ops.from_channels_last(y, Cout)

    @staticmethod
    def backward(ctx, dout):
        xcl, wd = ctx.saved_tensors
        d = ctx.d
        dy = dout.contiguous().float() if ctx.cl else ops.to_channels_last(dout.contiguous().float())
        dx = None
        if ctx.needs_input_grad[0] and streams.unit_helpers(dy):
            with streams.fork(dy.device, None, (dy, xcl)) as f:            # weight gradient beside the data gradient
                dw = ops.conv_wgrad(d, ops.view(xcl), dy)
            dx = ops.conv_dgrad(d, dy, wd)
            f.join(dw)
        else:
            dw = ops.conv_wgrad(d, ops.view(xcl), dy)
            if ctx.needs_input_grad[0]:
                dx = ops.conv_dgrad(d, dy, wd)
        if dx is not None and not ctx.cl:
            dx = ops.from_channels_last(dx, d.Cin)
        return dx, dw, None, None, None


def conv_plain(x, conv: torch.nn.Conv3d):
    if conv.bias is not None:
        raise NotImplementedError("mi355x hot path: plain convolution with bias is not used by the reference")
    if isinstance(x, CLAct):
        return CLAct(ConvFunction.apply(x.t, conv.weight, conv.stride, conv.padding, x.C), conv.weight.shape[0])
    return ConvFunction.apply(x, conv.weight, conv.stride, conv.padding)


class MaxPool1x3x3Function(torch.autograd.Function):
    """MaxPool3d((1,3,3), stride (1,2,2), padding (0,1,1))   (reference resnet.py:225)."""

    @staticmethod
    def forward(ctx, x):
        ops.require_cuda(x)
        x = ops.f32(x).contiguous()
        B, Cc, T, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty((B, Cc, T, Ho, Wo), device=x.device, dtype=torch.float32)
        idx = torch.empty((B, Cc, T, Ho, Wo), device=x.device, dtype=torch.int32)
        N.check(N.lib().md_maxpool_1x3x3_fwd(ops._p(x), B * Cc * T, H, W, ops._p(out), ops._p(idx), ops._stream()),
                "md_maxpool_1x3x3_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (B, Cc, T, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, Cc, T, H, W = ctx.shape
        g = ops.f32(dout).contiguous()
        dx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        N.check(N.lib().md_maxpool_1x3x3_bwd(ops._p(g), ops._p(idx), B * Cc * T, H, W, ops._p(dx), ops._stream()),
                "md_maxpool_1x3x3_bwd")
        return dx


class GlobalAvgPoolFunction(torch.autograd.Function):
    """F.adaptive_avg_pool3d(x, 1).view(-1, C)   (reference slowfast.py:33-34, 86-87)."""

    @staticmethod
    def forward(ctx, x):
        ops.require_cuda(x)
        x = ops.f32(x).contiguous()
        B, Cc = x.shape[0], x.shape[1]
        thw = x.numel() // (B * Cc)
        out = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
        N.check(N.lib().md_rowmean_fwd(ops._p(x), B * Cc, thw, ops._p(out), ops._stream()), "md_rowmean_fwd")
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        g = ops.f32(dout).contiguous()
        B, Cc = ctx.shape[0], ctx.shape[1]
        thw = 1
        for s in ctx.shape[2:]:
            thw *= s
        dx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        N.check(N.lib().md_rowmean_bwd(ops._p(g), B * Cc, thw, ops._p(dx), ops._stream()), "md_rowmean_bwd")
        return dx


class LSTMDirectionFunction(torch.autograd.Function):
    """One direction of one nn.LSTM layer with zero initial state: x (S,B,I) -> h (S,B,H)   (md_lstm_fwd / md_lstm_bwd)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, reverse):
        ops.require_cuda(x.contiguous(), w_ih, w_hh, b_ih, b_hh)
        x = ops.f32(x).contiguous()
        S, B, I = x.shape
        H = w_hh.shape[1]
        h = torch.empty((S, B, H), device=x.device, dtype=torch.float32)
        c = torch.empty_like(h)
        gates = torch.empty((S, B, 4 * H), device=x.device, dtype=torch.float32)
        N.check(N.lib().md_lstm_fwd(ops._p(x), ops._p(w_ih.contiguous()), ops._p(w_hh.contiguous()), ops._p(b_ih.contiguous()),
                                    ops._p(b_hh.contiguous()), S, B, I, H, int(bool(reverse)), ops._p(h), ops._p(c), ops._p(gates),
                                    ops._stream()), "md_lstm_fwd")
        ctx.save_for_backward(x, w_ih, w_hh, h, c, gates)
        ctx.reverse = int(bool(reverse))
        return h

    @staticmethod
    def backward(ctx, dh):
        x, w_ih, w_hh, h, c, gates = ctx.saved_tensors
        S, B, I = x.shape
        H = w_hh.shape[1]
        g = ops.f32(dh).contiguous()
        dx = torch.empty_like(x)
        dw_ih = torch.empty((4 * H, I), device=x.device); dw_hh = torch.empty((4 * H, H), device=x.device)
        db = torch.empty(4 * H, device=x.device)
        scratch = torch.empty(S * B * 4 * H, device=x.device)
        N.check(N.lib().md_lstm_bwd(ops._p(g), ops._p(x), ops._p(w_ih.contiguous()), ops._p(w_hh.contiguous()), ops._p(h),
                                    ops._p(c), ops._p(gates), S, B, I, H, ctx.reverse, ops._p(dx), ops._p(dw_ih), ops._p(dw_hh),
                                    ops._p(db), ops._p(scratch), ops._stream()), "md_lstm_bwd")
        return dx, dw_ih, dw_hh, db, db.clone(), None


class LSTMRecurrentFunction(torch.autograd.Function):
    """The recurrence of one LSTM direction on a precomputed input projection: xproj (S,B,4H) = x W_ih^T -> h (S,B,H), with W_hh
    held in registers for the whole sequence (md_lstm_rec_*; H = 64 / 128)."""

    @staticmethod
    def forward(ctx, xproj, w_hh, b_ih, b_hh, reverse):
        xproj = ops.f32(xproj).contiguous()
        ops.require_cuda(xproj, w_hh, b_ih, b_hh)
        S, B, H4 = xproj.shape
        H = H4 // 4
        h = torch.empty((S, B, H), device=xproj.device, dtype=torch.float32)
        c = torch.empty_like(h)
        gates = torch.empty_like(xproj)
        N.check(N.lib().md_lstm_rec_fwd(ops._p(xproj), ops._p(w_hh.contiguous()), ops._p(b_ih.contiguous()), ops._p(b_hh.contiguous()), S, B,
                                        H, int(bool(reverse)), ops._p(h), ops._p(c), ops._p(gates), ops._stream()), "md_lstm_rec_fwd")
        ctx.save_for_backward(w_hh, h, c, gates)
        ctx.reverse = int(bool(reverse))
        return h

    @staticmethod
    def backward(ctx, dh):
        w_hh, h, c, gates = ctx.saved_tensors
        S, B, H = h.shape
        dpre = torch.empty_like(gates)
        rows = S * B
        # many (t, b) rows (the scripts' batch 256: 5376): dW_hh = dpre^T h_prev is a weight-gradient GEMM for the matrix cores and
        # db a column sum; the one-thread-per-weight kernel took 5.3 ms per direction there (42 of CnnLSTM's 47 ms step)
        gemm = rows >= 512 and not N.lib().md_get_exact_fp32()
        dw_hh = None if gemm else torch.empty((4 * H, H), device=h.device)
        db = torch.empty(4 * H, device=h.device)
        N.check(N.lib().md_lstm_rec_bwd(ops._p(ops.f32(dh).contiguous()), ops._p(w_hh.contiguous()), ops._p(h), ops._p(c), ops._p(gates), S, B, H,
                                        ctx.reverse, ops._p(dpre), ops._p(dw_hh), None if gemm else ops._p(db), ops._stream()),
                "md_lstm_rec_bwd")
        if gemm:
            zero = h.new_zeros(1, B, H)
            hprev = torch.cat((h[1:], zero)) if ctx.reverse else torch.cat((zero, h[:-1]))       # h of the step processed before
            d = ops.make_desc(1, 1, 1, rows, H, 4 * H, (1, 1, 1), (1, 1, 1), (0, 0, 0))
            dw_hh = ops.conv_wgrad(d, ops.view(hprev.view(rows, H)), dpre.view(rows, 4 * H)).view(4 * H, H)
            ns = N.lib().md_channel_bias_bwd_scratch_floats(rows, 4 * H, 1)
            scratch = torch.empty(ns, device=h.device) if ns else None
            N.check(N.lib().md_channel_bias_bwd(ops._p(dpre), rows, 4 * H, 1, ops._p(db), ops._p(scratch), ops._stream()), "md_channel_bias_bwd")
        return dpre, dw_hh, db, db.clone(), None


import os as _os
_LSTM_BI = _os.environ.get("MD_LSTM_BI", "1") != "0"       # A/B switch: one launch per direction again


class LSTMBiRecurrentFunction(torch.autograd.Function):
    """Both directions of one bidirectional LSTM layer on their precomputed input projections, one launch each way
    (md_lstm_rec_fwd2 / md_lstm_rec_bwd2): (xproj_f, xproj_r) (S,B,4H) each -> (h_f, h_r).  Same arithmetic per direction as
    LSTMRecurrentFunction -- the two recurrences only stop waiting for each other."""

    @staticmethod
    def forward(ctx, xf, xr, whf, whr, bif, bhf, bir, bhr):
        xf = ops.f32(xf).contiguous(); xr = ops.f32(xr).contiguous()
        ops.require_cuda(xf, xr, whf, whr, bif, bhf, bir, bhr)
        S, B, H4 = xf.shape
        H = H4 // 4
        dev = xf.device
        h = [torch.empty((S, B, H), device=dev) for _ in range(2)]
        c = [torch.empty((S, B, H), device=dev) for _ in range(2)]
        gates = [torch.empty_like(xf), torch.empty_like(xr)]
        wh = [whf.contiguous(), whr.contiguous()]
        P = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])      # noqa: E731
        N.check(N.lib().md_lstm_rec_fwd2(P([xf, xr]), P(wh), P([bif.contiguous(), bir.contiguous()]), P([bhf.contiguous(), bhr.contiguous()]),
                                         S, B, H, P(h), P(c), P(gates), ops._stream()), "md_lstm_rec_fwd2")
        ctx.save_for_backward(wh[0], wh[1], h[0], h[1], c[0], c[1], gates[0], gates[1])
        return h[0], h[1]

    @staticmethod
    def backward(ctx, dhf, dhr):
        whf, whr, hf, hr, cf, cr, gf, gr = ctx.saved_tensors
        S, B, H = hf.shape
        dev = hf.device
        rows = S * B
        dh = [ops.f32(dhf if dhf is not None else torch.zeros_like(hf)).contiguous(), ops.f32(dhr if dhr is not None else torch.zeros_like(hr)).contiguous()]
        dpre = [torch.empty_like(gf), torch.empty_like(gr)]
        gemm = rows >= 512 and not N.lib().md_get_exact_fp32()
        dw = None if gemm else [torch.empty((4 * H, H), device=dev) for _ in range(2)]
        db = [torch.empty(4 * H, device=dev) for _ in range(2)]
        P = lambda ts: (C.c_void_p * 2)(*[t.data_ptr() for t in ts])      # noqa: E731
        N.check(N.lib().md_lstm_rec_bwd2(P(dh), P([whf, whr]), P([hf, hr]), P([cf, cr]), P([gf, gr]), S, B, H, P(dpre),
                                         None if gemm else P(dw), None if gemm else P(db), ops._stream()), "md_lstm_rec_bwd2")
        if gemm:        # many (t, b) rows: the MFMA weight gradient + column sums, per direction (see LSTMRecurrentFunction.backward)
            dw = []
            for d_, (h_, dp_) in enumerate(((hf, dpre[0]), (hr, dpre[1]))):
                zero = h_.new_zeros(1, B, H)
                hprev = torch.cat((h_[1:], zero)) if d_ else torch.cat((zero, h_[:-1]))
                dd = ops.make_desc(1, 1, 1, rows, H, 4 * H, (1, 1, 1), (1, 1, 1), (0, 0, 0))
                dw.append(ops.conv_wgrad(dd, ops.view(hprev.view(rows, H)), dp_.view(rows, 4 * H)).view(4 * H, H))
                ns = N.lib().md_channel_bias_bwd_scratch_floats(rows, 4 * H, 1)
                scratch = torch.empty(ns, device=dev) if ns else None
                N.check(N.lib().md_channel_bias_bwd(ops._p(dp_), rows, 4 * H, 1, ops._p(db[d_]), ops._p(scratch), ops._stream()), "md_channel_bias_bwd")
        return dpre[0], dpre[1], dw[0], dw[1], db[0], db[0].clone(), db[1], db[1].clone()


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one layer: register-resident recurrence behind an MFMA input projection where the width allows,
    else the general kernels."""
    H = w_hh.shape[1]
    if N.lib().md_lstm_rec_supported(int(H)):
        S, B, I = x.shape
        xproj = LinearRowsFunction.apply(x.reshape(S * B, I), w_ih).reshape(S, B, 4 * H)
        return LSTMRecurrentFunction.apply(xproj, w_hh, b_ih, b_hh, reverse)
    return LSTMDirectionFunction.apply(x, w_ih, w_hh, b_ih, b_hh, reverse)


def lstm_forward(x, lstm: torch.nn.LSTM):
    """nn.LSTM(batch_first=False, zero initial state, no dropout, no projection) on the gfx950 kernels: returns the output
    sequence (S, B, num_directions*H) -- what the reference's encoders consume (CnnLSTM.py:96-97)."""
    if lstm.batch_first or lstm.proj_size != 0 or not lstm.bias:
        raise NotImplementedError("mi355x hot path: this nn.LSTM configuration is not used by the reference")
    out = x
    for layer in range(lstm.num_layers):
        if layer > 0 and lstm.dropout > 0 and lstm.training:
            # nn.LSTM's inter-layer dropout: mask drawn with torch's device generator (no bit-compatibility with MIOpen's
            # internal dropout state is possible; the distribution is the same), applied by md_mask_scale
            keep = 1.0 - lstm.dropout
            mask = torch.empty_like(out).bernoulli_(keep)
            out = _MaskScale.apply(out, mask, 1.0 / keep)
        dirs = []
        if lstm.bidirectional and out.is_cuda and N.lib().md_lstm_rec_supported(int(lstm.hidden_size)) and _LSTM_BI:
            # both directions' recurrences in one launch each way (the input projections stay one MFMA GEMM per direction)
            S, B, I = out.shape
            par = [[getattr(lstm, n + f"_l{layer}" + sfx) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")] for sfx in ("", "_reverse")]
            xp = [LinearRowsFunction.apply(out.reshape(S * B, I), p[0]).reshape(S, B, 4 * lstm.hidden_size) for p in par]
            dirs = list(LSTMBiRecurrentFunction.apply(xp[0], xp[1], par[0][1], par[1][1], par[0][2], par[0][3], par[1][2], par[1][3]))
        else:
            for rev in range(2 if lstm.bidirectional else 1):
                sfx = f"_l{layer}" + ("_reverse" if rev else "")
                dirs.append(lstm_direction(out, getattr(lstm, "weight_ih" + sfx), getattr(lstm, "weight_hh" + sfx),
                                           getattr(lstm, "bias_ih" + sfx), getattr(lstm, "bias_hh" + sfx), rev))
        out = dirs[0] if len(dirs) == 1 else torch.cat(dirs, dim=2)
    return out


class _ChannelBias(torch.autograd.Function):
    """x (N, C, L) + bias[c]   (md_channel_bias_*)."""
    @staticmethod
    def forward(ctx, x, bias):                      # x (N, C, L)
        x = ops.f32(x).contiguous()
        out = torch.empty_like(x)
        Nn, Cc, L = x.shape
        N.check(N.lib().md_channel_bias_fwd(ops._p(x), ops._p(bias.contiguous()), Nn, Cc, L, ops._p(out), ops._stream()),
                "md_channel_bias_fwd")
        ctx.shape = (Nn, Cc, L)
        return out

    @staticmethod
    def backward(ctx, dout):
        Nn, Cc, L = ctx.shape
        g = ops.f32(dout).contiguous()
        db = torch.empty(Cc, device=g.device)
        ns = N.lib().md_channel_bias_bwd_scratch_floats(Nn, Cc, L)
        scratch = torch.empty(ns, device=g.device) if ns else None
        N.check(N.lib().md_channel_bias_bwd(ops._p(g), Nn, Cc, L, ops._p(db), ops._p(scratch), ops._stream()), "md_channel_bias_bwd")
        return g, db


class PatchEmbedFunction(torch.autograd.Function):
    """ViViT's patch embedding + space token + positional table as one gather-GEMM (md_patch_embed_*; reference ViViT.py:141-148,
    175-184).  x: the clip as a (b, t, c, H, W) tensor or view with contiguous image rows (read in place through its strides);
    w_perm (dim, c*p*p): the Linear weight with columns in (c, p1, p2) order; pos (t, n+1, dim); token (dim).
    Returns (b*t, n+1, dim).  The clip's own gradient is only produced when asked for (it is the model input: training never
    asks): patch gradients from the MFMA Linear, scattered back with a view permutation."""
    @staticmethod
    def forward(ctx, x, w_perm, bias, pos, token, patch):
        ops.require_cuda(w_perm, bias, pos, token)
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 5 or x.stride(4) != 1 or x.stride(3) != x.shape[4]:
            raise RuntimeError("mi355x hot path: patch embedding expects a CUDA fp32 (b, t, c, H, W) clip with contiguous image rows")
        b, t, c, H, W = x.shape
        ops.check_fp16_range(x, "the clip")
        dim = w_perm.shape[0]
        n = (H // patch) * (W // patch)
        out = torch.empty((b * t, n + 1, dim), device=x.device, dtype=torch.float32)
        geo = (b, t, c, H, W, x.stride(0), x.stride(1), x.stride(2), int(patch))
        N.check(N.lib().md_patch_embed_fwd(C.c_void_p(x.data_ptr()), b, t, c, H, W, x.stride(0), x.stride(1), x.stride(2), int(patch),
                                           ops._p(w_perm), ops._p(bias), ops._p(pos), ops._p(token), dim, ops._p(out), ops._stream()),
                "md_patch_embed_fwd")
        ctx.geo, ctx.dim, ctx.n = geo, dim, n
        ctx.save_for_backward(x, w_perm)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w_perm = ctx.saved_tensors
        b, t, c, H, W, sb, st, sc, patch = ctx.geo
        dim, n = ctx.dim, ctx.n
        g = ops.f32(dout).contiguous()                                          # (b*t, n+1, dim)
        L = N.lib()
        dw = torch.empty((dim, c * patch * patch), device=g.device, dtype=torch.float32)
        ws = torch.empty(L.md_patch_embed_wgrad_workspace_floats(b, t, c, H, W, patch, dim), device=g.device, dtype=torch.float32)
        N.check(L.md_patch_embed_wgrad(C.c_void_p(x.data_ptr()), b, t, c, H, W, sb, st, sc, patch, ops._p(g), dim, ops._p(dw), ops._p(ws),
                                       ops._stream()), "md_patch_embed_wgrad")
        # positional table: column sums over the batch (fixed-order two-level reduction); bias and token follow from it
        Cc = t * (n + 1) * dim
        dpos = torch.empty(Cc, device=g.device, dtype=torch.float32)
        ns = L.md_channel_bias_bwd_scratch_floats(b, Cc, 1)
        scratch = torch.empty(ns, device=g.device) if ns else None
        N.check(L.md_channel_bias_bwd(ops._p(g), b, Cc, 1, ops._p(dpos), ops._p(scratch), ops._stream()), "md_channel_bias_bwd")
        dpos = dpos.view(t, n + 1, dim)
        dbias = dpos[:, 1:].sum(dim=(0, 1))
        dtoken = dpos[:, 0].sum(dim=0)
        dx = None
        if ctx.needs_input_grad[0]:
            with torch.no_grad():
                dp = LinearRowsFunction.apply(g[:, 1:].reshape(b * t * n, dim), w_perm.t().contiguous())      # (M, c*p*p), columns (c p1 p2)
            nh, nw = H // patch, W // patch
            dx = dp.view(b, t, nh, nw, c, patch, patch).permute(0, 1, 4, 2, 5, 3, 6).reshape(b, t, c, H, W)
        return dx, dw, dbias, dpos, dtoken, None


class _SeqSum(torch.autograd.Function):
    """scale * sum over the sequence axis of x (B, S, D)   (md_seq_sum_*): the closed form of the reference's attention
    pooling (see models/CnnLSTM.py); extra parameters passed in get exact zero gradients."""
    @staticmethod
    def forward(ctx, x, scale, *zero_grad_params):   # x (B, S, D) -> (B, D); the extra parameters get exact zero gradients
        x = ops.f32(x).contiguous()
        B, S, D = x.shape
        out = torch.empty((B, D), device=x.device)
        N.check(N.lib().md_seq_sum_fwd(ops._p(x), B, S, D, float(scale), ops._p(out), ops._stream()), "md_seq_sum_fwd")
        ctx.shape = (B, S, D); ctx.scale = float(scale)
        ctx.zshapes = [tuple(p.shape) for p in zero_grad_params]
        return out

    @staticmethod
    def backward(ctx, dout):
        B, S, D = ctx.shape
        g = ops.f32(dout).contiguous()
        dx = torch.empty((B, S, D), device=g.device)
        N.check(N.lib().md_seq_sum_bwd(ops._p(g), B, S, D, ctx.scale, ops._p(dx), ops._stream()), "md_seq_sum_bwd")
        return (dx, None) + tuple(torch.zeros(s, device=g.device) for s in ctx.zshapes)


class _AbsorbedBias(torch.autograd.Function):
    """A convolution / linear bias in front of a training-mode BatchNorm cancels in the output; its gradient is exactly zero."""
    @staticmethod
    def forward(ctx, out, bias):
        ctx.n = bias.numel()
        return out.view_as(out)

    @staticmethod
    def backward(ctx, dout):
        return dout, torch.zeros(ctx.n, device=dout.device, dtype=dout.dtype)


class _MaskScale(torch.autograd.Function):
    """x * mask * scale (inverted dropout with a given mask; md_mask_scale)."""

    @staticmethod
    def forward(ctx, x, mask, scale):
        x = ops.f32(x).contiguous()
        out = torch.empty_like(x)
        N.check(N.lib().md_mask_scale(ops._p(x), ops._p(mask), float(scale), x.numel(), ops._p(out), ops._stream()), "md_mask_scale")
        ctx.save_for_backward(mask); ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        (mask,) = ctx.saved_tensors
        g = ops.f32(dout).contiguous()
        dx = torch.empty_like(g)
        N.check(N.lib().md_mask_scale(ops._p(g), ops._p(mask), ctx.scale, g.numel(), ops._p(dx), ops._stream()), "md_mask_scale")
        return dx, None, None


def _unit_with_bias(x5, w5, bias, bn, stride, padding, slope: float, training: bool):
    b = None if bias is None else bias.detach()
    rmean = bn.running_mean if (training or b is None) else bn.running_mean - b
    z = ConvBnLeakyFunction.apply(x5, w5, bn.weight, bn.bias, rmean, bn.running_var, stride, padding, float(slope), bool(training),
                                  float(bn.eps), float(bn.momentum))
    if training:
        if b is not None:
            bn.running_mean.add_(b * bn.momentum)
            z = _AbsorbedBias.apply(z, bias)
        bump_batches_tracked(bn)
    return z


def conv1d_bn_leaky(x_bct, conv: torch.nn.Conv1d, bn: torch.nn.BatchNorm1d, slope: float, training: bool):
    """Conv1d (with or without bias) -> BatchNorm1d -> LeakyReLU(slope) on (B, C, T) as a (k,1,1) unit of the conv kernels;
    a bias only shifts the batch mean, so it is folded into the running mean (state dicts stay interchangeable with the
    reference) and gets its exact zero gradient."""
    z = _unit_with_bias(x_bct.contiguous()[:, :, :, None, None], conv.weight[:, :, :, None, None], conv.bias, bn,
                        (conv.stride[0], 1, 1), (conv.padding[0], 0, 0), slope, training)
    return z.squeeze(4).squeeze(3)          # (views: a select here would cost zeros + copy in the backward)


def linear_bn_leaky(x, lin: torch.nn.Linear, bn: torch.nn.BatchNorm1d, slope: float, training: bool):
    """Linear -> BatchNorm1d -> LeakyReLU(slope) (slope 0 = ReLU) on (B, D) rows: the same unit with a 1x1x1 kernel."""
    z = _unit_with_bias(x.contiguous()[:, :, None, None, None], lin.weight[:, :, None, None, None], lin.bias, bn, (1, 1, 1), (0, 0, 0),
                        slope, training)
    return z[:, :, 0, 0, 0]


class OuterFusionFunction(torch.autograd.Function):
    """[1 | a] (x) [1 | c] per sample, flattened: (B, Da), (B, Dc) -> (B, (Da+1)(Dc+1))   (md_outer_*)."""

    @staticmethod
    def forward(ctx, a, c):
        a = ops.f32(a).contiguous(); c = ops.f32(c).contiguous()
        ops.require_cuda(a, c)
        B, Da = a.shape
        Dc = c.shape[1]
        out = torch.empty((B, (Da + 1) * (Dc + 1)), device=a.device)
        N.check(N.lib().md_outer_fwd(ops._p(a), ops._p(c), B, Da, Dc, ops._p(out), ops._stream()), "md_outer_fwd")
        ctx.save_for_backward(a, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, c = ctx.saved_tensors
        da = torch.empty_like(a); dc = torch.empty_like(c)
        N.check(N.lib().md_outer_bwd(ops._p(a), ops._p(c), ops._p(ops.f32(dout).contiguous()), a.shape[0], a.shape[1], c.shape[1],
                                     ops._p(da), ops._p(dc), ops._stream()), "md_outer_bwd")
        return da, dc


def linear(x, lin: torch.nn.Linear):
    """nn.Linear on (B, D) rows as a 1x1x1 convolution plus the per-channel bias kernel."""
    y = ConvFunction.apply(x.contiguous()[:, :, None, None, None], lin.weight[:, :, None, None, None], (1, 1, 1), (0, 0, 0))
    y = y[:, :, 0, 0, 0]
    return y if lin.bias is None else _ChannelBias.apply(y[:, :, None], lin.bias).squeeze(2)


def _ln_backward(g, gamma, xhat, rstd, dres):
    D = g.shape[-1]
    rows = g.numel() // D
    dx = torch.empty_like(g); dgamma = torch.empty(D, device=g.device); dbeta = torch.empty(D, device=g.device)
    ns = N.lib().md_add_layernorm_bwd_scratch_floats(rows, D)
    scratch = torch.empty(ns, device=g.device) if ns else None
    N.check(N.lib().md_add_layernorm_bwd(ops._p(g), ops._p(gamma.contiguous()), ops._p(xhat), ops._p(rstd), ops._p(dres), rows, D,
                                         ops._p(dx), ops._p(dgamma), ops._p(dbeta), ops._p(scratch), ops._stream()),
            "md_add_layernorm_bwd")
    return dx, dgamma, dbeta


class AddLayerNormFunction(torch.autograd.Function):
    """LayerNorm(a + b) over the last dimension (b may be None)   (md_add_layernorm_*)."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps):
        a = ops.f32(a).contiguous()
        ops.require_cuda(a, gamma, beta)
        D = a.shape[-1]
        rows = a.numel() // D
        bb = None if b is None else ops.f32(b).contiguous()
        out = torch.empty_like(a); xhat = torch.empty_like(a); rstd = torch.empty(rows, device=a.device)
        N.check(N.lib().md_add_layernorm_fwd(ops._p(a), ops._p(bb), ops._p(gamma.contiguous()), ops._p(beta.contiguous()), rows, D,
                                             float(eps), ops._p(out), ops._p(xhat), ops._p(rstd), None, ops._stream()),
                "md_add_layernorm_fwd")
        ctx.save_for_backward(gamma, xhat, rstd)
        ctx.has_b = b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        gamma, xhat, rstd = ctx.saved_tensors
        dx, dgamma, dbeta = _ln_backward(ops.f32(dout).contiguous(), gamma, xhat, rstd, None)
        return dx, (dx if ctx.has_b else None), dgamma, dbeta, None


class ResidualLayerNormFunction(torch.autograd.Function):
    """(s, h) = (a + b, LayerNorm(a + b)): one step of a pre-norm residual stream (reference ViViT.py:108-111).  The backward adds
    the gradient that arrives through s to the LayerNorm's, so the stream never fans out in autograd."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps):
        a = ops.f32(a).contiguous(); b = ops.f32(b).contiguous()
        ops.require_cuda(a, b, gamma, beta)
        ctx.set_materialize_grads(False)
        D = a.shape[-1]
        rows = a.numel() // D
        out = torch.empty_like(a); xhat = torch.empty_like(a); rstd = torch.empty(rows, device=a.device); s = torch.empty_like(a)
        N.check(N.lib().md_add_layernorm_fwd(ops._p(a), ops._p(b), ops._p(gamma.contiguous()), ops._p(beta.contiguous()), rows, D,
                                             float(eps), ops._p(out), ops._p(xhat), ops._p(rstd), ops._p(s), ops._stream()),
                "md_add_layernorm_fwd")
        ctx.save_for_backward(gamma, xhat, rstd)
        return s, out

    @staticmethod
    def backward(ctx, ds, dout):
        gamma, xhat, rstd = ctx.saved_tensors
        if dout is None:                                  # the normalised branch was not used: identity for the stream
            return ds, ds, torch.zeros_like(gamma), torch.zeros_like(gamma), None
        dres = None if ds is None else ops.f32(ds).contiguous()
        dx, dgamma, dbeta = _ln_backward(ops.f32(dout).contiguous(), gamma, xhat, rstd, dres)
        return dx, dx, dgamma, dbeta, None


class BranchResidualLayerNormFunction(torch.autograd.Function):
    """(s, h) = (dropout(y + bias) + res, LayerNorm(s)): ResidualLayerNormFunction with the branch's Linear bias and nn.Dropout folded
    into the same pass (md_branch_layernorm_*; reference ViViT.py:31-46,85-91,108-111).  y is the Linear's raw product; ``site`` =
    (key tensor, tag) of the dropout call site (ops.dropout_site()) or None for no dropout."""

    @staticmethod
    def forward(ctx, y, bias, res, gamma, beta, eps, site, keep):
        y = ops.f32(y).contiguous(); res = ops.f32(res).contiguous()
        ops.require_cuda(y, bias, res, gamma, beta)
        ctx.set_materialize_grads(False)
        D = y.shape[-1]
        rows = y.numel() // D
        bb = None if bias is None else ops.f32(bias).contiguous()
        key, tag = site if site is not None else (None, 0)
        out = torch.empty_like(y); xhat = torch.empty_like(y); rstd = torch.empty(rows, device=y.device); s = torch.empty_like(y)
        N.check(N.lib().md_branch_layernorm_fwd(ops._p(y), ops._p(bb), ops._p(key), int(tag), float(keep), ops._p(res), ops._p(gamma.contiguous()),
                                                ops._p(beta.contiguous()), rows, D, float(eps), ops._p(out), ops._p(xhat), ops._p(rstd), ops._p(s),
                                                ops._stream()), "md_branch_layernorm_fwd")
        ctx.save_for_backward(gamma, xhat, rstd)
        ctx.key, ctx.tag, ctx.keep, ctx.has_bias = key, int(tag), float(keep), bias is not None
        return s, out

    @staticmethod
    def backward(ctx, ds, dout):
        gamma, xhat, rstd = ctx.saved_tensors
        D = xhat.shape[-1]
        rows = xhat.numel() // D
        dev = xhat.device
        if dout is None:
            # the normalised branch was not used: only the stream's gradient flows; the branch's is that gradient through the dropout
            dsum = ops.f32(ds).contiguous()
            if ctx.key is None:
                dy = dsum
            else:
                dy = torch.empty_like(dsum)
                N.check(N.lib().md_dropout_ctr(ops._p(dsum), ops._p(ctx.key), ctx.tag, ctx.keep, 1.0 / ctx.keep, dsum.numel(), ops._p(dy),
                                               ops._stream()), "md_dropout_ctr")
            dgamma, dbeta = torch.zeros_like(gamma), torch.zeros_like(gamma)
        else:
            g = ops.f32(dout).contiguous()
            dres = None if ds is None else ops.f32(ds).contiguous()
            dsum = torch.empty_like(g); dy = torch.empty_like(g)
            dgamma = torch.empty(D, device=dev); dbeta = torch.empty(D, device=dev)
            db = torch.empty(D, device=dev) if ctx.has_bias else None
            scratch = torch.empty(N.lib().md_branch_layernorm_bwd_scratch_floats(rows, D), device=dev)
            N.check(N.lib().md_branch_layernorm_bwd(ops._p(g), ops._p(gamma.contiguous()), ops._p(xhat), ops._p(rstd), ops._p(dres), ops._p(ctx.key),
                                                    ctx.tag, ctx.keep, rows, D, ops._p(dsum), ops._p(dy), ops._p(dgamma), ops._p(dbeta),
                                                    ops._p(db), ops._p(scratch), ops._stream()), "md_branch_layernorm_bwd")
            return dy, db, dsum, dgamma, dbeta, None, None, None
        db = None
        if ctx.has_bias:
            db = torch.empty(D, device=dev)
            ns = N.lib().md_channel_bias_bwd_scratch_floats(rows, D, 1)
            scratch = torch.empty(ns, device=dev) if ns else None
            N.check(N.lib().md_channel_bias_bwd(ops._p(dy), rows, D, 1, ops._p(db), ops._p(scratch), ops._stream()), "md_channel_bias_bwd")
        return dy, db, dsum, dgamma, dbeta, None, None, None


def branch_residual_layernorm(y2d, bias, p: float, training: bool, res, norm):
    """One residual step of a pre-norm transformer block whose branch ends in Linear (raw product y2d, its bias) -> Dropout(p):
    returns (new stream, LayerNorm(new stream)) shaped like ``res``.  Fused where md_branch_layernorm_* applies, composed otherwise."""
    D = res.shape[-1]
    rows = res.numel() // D
    drop = training and p > 0.0
    site = ops.dropout_site() if (drop and res.is_cuda) else None
    if res.is_cuda and N.lib().md_branch_layernorm_supported(rows, D) and (not drop or site is not None):
        s, h = BranchResidualLayerNormFunction.apply(y2d.reshape(res.shape), bias, res, norm.weight, norm.bias, norm.eps, site, 1.0 - p if drop else 1.0)
        return s, h
    y = y2d if bias is None else _ChannelBias.apply(y2d.contiguous()[:, :, None], bias).squeeze(2)
    if drop:
        keep = 1.0 - p
        y = CtrDropoutFunction.apply(y, site[0], site[1], keep) if site is not None else \
            _MaskScale.apply(y, torch.empty_like(y).bernoulli_(keep), 1.0 / keep)
    return ResidualLayerNormFunction.apply(y.reshape(res.shape), res, norm.weight, norm.bias, norm.eps)


class AttentionFunction(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh) + mask) [* dropout] v per head   (md_attention_*).  qkv (S, B, 3D) -> (S, B, D), or with
    batch_first (B, S, 3D) -> (B, S, D)."""

    @staticmethod
    def forward(ctx, qkv, mask, heads, drop, batch_first=False):
        qkv = ops.f32(qkv).contiguous()
        ops.require_cuda(qkv, mask, drop)
        (B, S, D3) = qkv.shape if batch_first else (qkv.shape[1], qkv.shape[0], qkv.shape[2])
        D = D3 // 3
        ctx.heads = int(heads); ctx.has_drop = drop is not None; ctx.bf = bool(batch_first)
        ctx.lse = mask is None and drop is None and bool(N.lib().md_attention_lse_supported(S, D, int(heads)))
        if ctx.lse:        # no S x S matrices: one log-sum-exp per query row, the backward recomputes the probabilities
            lse = torch.empty((B * heads, S), device=qkv.device); out = torch.empty(qkv.shape[:2] + (D,), device=qkv.device)
            N.check(N.lib().md_attention_lse_fwd(ops._p(qkv), S, B, D, int(heads), int(batch_first), ops._p(lse), ops._p(out),
                                                 ops._stream()), "md_attention_lse_fwd")
            ctx.save_for_backward(qkv, lse)
            return out
        probs = torch.empty((B * heads, S, S), device=qkv.device); out = torch.empty(qkv.shape[:2] + (D,), device=qkv.device)
        N.check(N.lib().md_attention_fwd(ops._p(qkv), ops._p(mask), ops._p(drop), S, B, D, int(heads), int(batch_first), ops._p(probs),
                                         ops._p(out), ops._stream()), "md_attention_fwd")
        ctx.save_for_backward(qkv, probs, drop) if drop is not None else ctx.save_for_backward(qkv, probs)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.lse:
            qkv, lse = ctx.saved_tensors
            (B, S, D3) = qkv.shape if ctx.bf else (qkv.shape[1], qkv.shape[0], qkv.shape[2])
            g = ops.f32(dout).contiguous()
            dqkv = torch.empty_like(qkv)
            delta = torch.empty_like(lse)
            N.check(N.lib().md_attention_lse_bwd(ops._p(qkv), ops._p(lse), ops._p(g), S, B, D3 // 3, ctx.heads, int(ctx.bf), ops._p(dqkv),
                                                 ops._p(delta), ops._stream()), "md_attention_lse_bwd")
            return dqkv, None, None, None, None
        if ctx.has_drop:
            qkv, probs, drop = ctx.saved_tensors
        else:
            (qkv, probs), drop = ctx.saved_tensors, None
        (B, S, D3) = qkv.shape if ctx.bf else (qkv.shape[1], qkv.shape[0], qkv.shape[2])
        g = ops.f32(dout).contiguous()
        dqkv = torch.empty_like(qkv)
        scratch = torch.empty_like(probs)
        N.check(N.lib().md_attention_bwd(ops._p(qkv), ops._p(probs), ops._p(drop), ops._p(g), S, B, D3 // 3, ctx.heads, int(ctx.bf),
                                         ops._p(dqkv), ops._p(scratch), ops._stream()), "md_attention_bwd")
        return dqkv, None, None, None, None


class EluFunction(torch.autograd.Function):
    """nn.ELU(alpha)   (md_elu)."""

    @staticmethod
    def forward(ctx, x, alpha):
        x = ops.f32(x).contiguous()
        ops.require_cuda(x)
        out = torch.empty_like(x)
        N.check(N.lib().md_elu(ops._p(x), None, float(alpha), x.numel(), ops._p(out), ops._stream()), "md_elu")
        ctx.save_for_backward(x); ctx.alpha = float(alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        N.check(N.lib().md_elu(ops._p(x), ops._p(ops.f32(dout).contiguous()), ctx.alpha, x.numel(), ops._p(dx), ops._stream()), "md_elu")
        return dx, None


class LinearRowsFunction(torch.autograd.Function):
    """x (rows, Din) @ w (Dout, Din)^T with the rows laid along the W axis of one 1x1x1 convolution (row-major rows ARE the
    channels-last layout), so 128-row tiles go through the MFMA kernels however many rows there are."""

    @staticmethod
    def forward(ctx, x, w):
        x = ops.f32(x).contiguous(); w = ops.f32(w).contiguous()
        ops.require_cuda(x, w)
        ops.check_fp16_range(x, "a Linear's input")
        rows, Din = x.shape
        Dout = w.shape[0]
        d = ops.make_desc(1, 1, 1, rows, Din, Dout, (1, 1, 1), (1, 1, 1), (0, 0, 0))
        if Din % 4:
            x = torch.nn.functional.pad(x, (0, ops.cpad(Din) - Din))
        wf, wd = ops.pack_weights(d, w[:, :, None, None, None].contiguous(), want_dgrad=True, owner=w)
        y, _ = ops.conv_fwd(d, ops.view(x), wf, x.device, want_stats=False)
        ctx.d = d
        ctx.save_for_backward(x, wd)
        return y.view(rows, ops.cpad(Dout))[:, :Dout]

    @staticmethod
    def backward(ctx, dout):
        x, wd = ctx.saved_tensors
        d = ctx.d
        dy = ops.f32(dout)
        if d.Cout % 4:
            dy = torch.nn.functional.pad(dy, (0, ops.cpad(d.Cout) - d.Cout))
        dy = dy.contiguous()
        dw = ops.conv_wgrad(d, ops.view(x), dy).view(d.Cout, d.Cin)
        dx = ops.conv_dgrad(d, dy, wd).view(-1, ops.cpad(d.Cin))[:, :d.Cin] if ctx.needs_input_grad[0] else None
        return dx, dw


class GeluFunction(torch.autograd.Function):
    """kind 0: nn.GELU (erf); kind 1: the reference's tanh form (transformer.py:35-37)   (md_gelu)."""

    @staticmethod
    def forward(ctx, x, kind):
        x = ops.f32(x).contiguous()
        ops.require_cuda(x)
        out = torch.empty_like(x)
        N.check(N.lib().md_gelu(ops._p(x), None, int(kind), x.numel(), ops._p(out), ops._stream()), "md_gelu")
        ctx.save_for_backward(x); ctx.kind = int(kind)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        g = ops.f32(dout).contiguous()
        dx = torch.empty_like(x)
        N.check(N.lib().md_gelu(ops._p(x), ops._p(g), ctx.kind, x.numel(), ops._p(dx), ops._stream()), "md_gelu")
        return dx, None


class BiasGeluDropFunction(torch.autograd.Function):
    """gelu(x + bias) * mask * scale over rows (md_bias_gelu_drop): the three elementwise passes between a FeedForward's two Linears
    in one, bit-identical to them.  mask None: no dropout."""

    @staticmethod
    def forward(ctx, x, bias, mask, scale, kind):
        x = ops.f32(x).contiguous(); bias = ops.f32(bias).contiguous()
        ops.require_cuda(x, bias, mask)
        rows, Cc = x.shape
        out = torch.empty_like(x)
        N.check(N.lib().md_bias_gelu_drop(ops._p(x), ops._p(bias), ops._p(mask), None, float(scale), int(kind), rows, Cc, ops._p(out),
                                          ops._stream()), "md_bias_gelu_drop")
        ctx.save_for_backward(x, bias) if mask is None else ctx.save_for_backward(x, bias, mask)
        ctx.scale, ctx.kind, ctx.has_mask = float(scale), int(kind), mask is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.has_mask:
            x, bias, mask = ctx.saved_tensors
        else:
            (x, bias), mask = ctx.saved_tensors, None
        g = ops.f32(dout).contiguous()
        rows, Cc = x.shape
        dx = torch.empty_like(x)
        N.check(N.lib().md_bias_gelu_drop(ops._p(x), ops._p(bias), ops._p(mask), ops._p(g), ctx.scale, ctx.kind, rows, Cc, ops._p(dx),
                                          ops._stream()), "md_bias_gelu_drop")
        db = torch.empty(Cc, device=g.device)
        ns = N.lib().md_channel_bias_bwd_scratch_floats(rows, Cc, 1)
        scratch = torch.empty(ns, device=g.device) if ns else None
        N.check(N.lib().md_channel_bias_bwd(ops._p(dx), rows, Cc, 1, ops._p(db), ops._p(scratch), ops._stream()), "md_channel_bias_bwd")
        return dx, db, None, None, None


class CtrDropoutFunction(torch.autograd.Function):
    """Inverted dropout whose decisions are regenerated, not stored (md_dropout_ctr): backward = the same call on the gradient."""

    @staticmethod
    def forward(ctx, x, state, tag, keep):
        x = ops.f32(x).contiguous()
        ops.require_cuda(x, state)
        out = torch.empty_like(x)
        N.check(N.lib().md_dropout_ctr(ops._p(x), ops._p(state), int(tag), float(keep), 1.0 / float(keep), x.numel(), ops._p(out), ops._stream()),
                "md_dropout_ctr")
        ctx.state, ctx.tag, ctx.keep = state, int(tag), float(keep)
        return out

    @staticmethod
    def backward(ctx, dout):
        g = ops.f32(dout).contiguous()
        dx = torch.empty_like(g)
        N.check(N.lib().md_dropout_ctr(ops._p(g), ops._p(ctx.state), ctx.tag, ctx.keep, 1.0 / ctx.keep, g.numel(), ops._p(dx), ops._stream()),
                "md_dropout_ctr")
        return dx, None, None, None


class BiasGeluDropCtrFunction(torch.autograd.Function):
    """gelu(x + bias) * keep-decision * scale over rows with the decisions regenerated (md_bias_gelu_drop_ctr)."""

    @staticmethod
    def forward(ctx, x, bias, state, tag, keep, kind):
        x = ops.f32(x).contiguous(); bias = ops.f32(bias).contiguous()
        ops.require_cuda(x, bias, state)
        rows, Cc = x.shape
        out = torch.empty_like(x)
        N.check(N.lib().md_bias_gelu_drop_ctr(ops._p(x), ops._p(bias), ops._p(state), int(tag), float(keep), None, 1.0 / float(keep), int(kind),
                                              rows, Cc, ops._p(out), ops._stream()), "md_bias_gelu_drop_ctr")
        ctx.save_for_backward(x, bias)
        ctx.state, ctx.tag, ctx.keep, ctx.kind = state, int(tag), float(keep), int(kind)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, bias = ctx.saved_tensors
        g = ops.f32(dout).contiguous()
        rows, Cc = x.shape
        dx = torch.empty_like(x)
        N.check(N.lib().md_bias_gelu_drop_ctr(ops._p(x), ops._p(bias), ops._p(ctx.state), ctx.tag, ctx.keep, ops._p(g), 1.0 / ctx.keep, ctx.kind,
                                              rows, Cc, ops._p(dx), ops._stream()), "md_bias_gelu_drop_ctr")
        db = torch.empty(Cc, device=g.device)
        ns = N.lib().md_channel_bias_bwd_scratch_floats(rows, Cc, 1)
        scratch = torch.empty(ns, device=g.device) if ns else None
        N.check(N.lib().md_channel_bias_bwd(ops._p(dx), rows, Cc, 1, ops._p(db), ops._p(scratch), ops._stream()), "md_channel_bias_bwd")
        return dx, db, None, None, None, None


def linear_bias_gelu_dropout(x2d, weight, bias, p: float, training: bool, kind: int = 0):
    """dropout(gelu(x @ weight^T + bias)): the MFMA Linear, then one fused elementwise pass (three separately when the width is not a
    multiple of 4 or the Linear has no bias)."""
    y = LinearRowsFunction.apply(x2d, weight)
    if bias is None or y.shape[1] % 4:
        return dropout(GeluFunction.apply(y if bias is None else _ChannelBias.apply(y.contiguous()[:, :, None], bias).squeeze(2), kind), p, training)
    y = y.contiguous()
    if training and p > 0.0:
        keep = 1.0 - p
        site = ops.dropout_site() if y.is_cuda else None
        if site is not None:
            return BiasGeluDropCtrFunction.apply(y, bias, site[0], site[1], keep, kind)
        return BiasGeluDropFunction.apply(y, bias, torch.empty_like(y).bernoulli_(keep), 1.0 / keep, kind)
    return BiasGeluDropFunction.apply(y, bias, None, 1.0, kind)


def linear_wb(x2d, weight, bias):
    """x (rows, D_in) @ weight(D_out, D_in)^T + bias: the rows-major 1x1x1 convolution plus the per-channel bias kernel."""
    y = LinearRowsFunction.apply(x2d, weight)
    return y if bias is None else _ChannelBias.apply(y.contiguous()[:, :, None], bias).squeeze(2)


def dropout(x, p: float, training: bool):
    """Inverted dropout.  Inside a model's ``ops.counter_dropout`` scope: mask-free (md_dropout_ctr); otherwise a mask from torch's
    device generator, applied by md_mask_scale."""
    if not training or p <= 0.0:
        return x
    keep = 1.0 - p
    site = ops.dropout_site() if x.is_cuda else None
    if site is not None:
        return CtrDropoutFunction.apply(x, site[0], site[1], keep)
    return _MaskScale.apply(x, torch.empty_like(x).bernoulli_(keep), 1.0 / keep)


def head_apply(f, lin0: torch.nn.Linear, bn: torch.nn.BatchNorm1d, lin1: torch.nn.Linear, alpha: float, training: bool):
    """Classifier head Linear -> BatchNorm1d -> activation -> Linear; alpha >= 0: ELU(alpha) (0 = ReLU), alpha < 0:
    LeakyReLU(-alpha).  The fused single-workgroup kernels (md_head_*) keep two (B, hidden) tiles in LDS; where those do not
    fit (MLSTM_FCN's 512 -> 256 head at batch 32, BASELINE configs[0]) the head is composed from the Linear+BatchNorm1d unit,
    md_elu and the MFMA Linear instead."""
    B, Hd = f.shape[0], lin0.out_features
    # wide heads (SlowFast: 640 -> 320): the fused kernels are ONE workgroup walking the whole weight matrix (350 us per pass at
    # 640 x 320); above 64k weights the composed form (MFMA Linear kernels over all CUs) is the faster one.  MD_HEAD_FUSED_MAX overrides.
    import os
    fused_max = int(os.environ.get("MD_HEAD_FUSED_MAX", "65536"))
    if 2 * B * Hd * 4 <= 60000 and lin0.in_features * Hd <= fused_max:
        out = HeadFunction.apply(f, lin0.weight, lin0.bias, bn.weight, bn.bias, lin1.weight, lin1.bias, bn.running_mean, bn.running_var,
                                 float(alpha), float(bn.eps), float(bn.momentum), bool(training))
        if training:
            bump_batches_tracked(bn)
        return out
    if alpha < 0:
        h = linear_bn_leaky(f, lin0, bn, -float(alpha), bool(training))
    else:
        h = EluFunction.apply(linear_bn_leaky(f, lin0, bn, 1.0, bool(training)), float(alpha))
    return linear_wb(h, lin1.weight, lin1.bias)

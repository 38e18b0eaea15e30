"""Thin torch-level wrappers over the C ABI (include/mi355x_disrupt.h).

PyTorch is used for device memory and the current HIP stream only; every function here launches
hand-written gfx950 kernels through ``_native.lib()`` and raises if given a CPU tensor.
Tensors in the library's internal layout are channels-last ``[N, T, H, W, Cp]`` fp32 with
``Cp = cpad(C)``.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional, Sequence, Tuple

import torch

from . import _native as N


def set_exact_fp32(on: bool) -> bool:
    """Process-wide arithmetic mode of the convolutions (see md_set_exact_fp32); returns the previous mode."""
    return bool(N.lib().md_set_exact_fp32(int(bool(on))))


def cpad(c: int) -> int:
    return (c + 3) & ~3


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def require_cuda(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mi355x hot path: CPU tensor given; this path runs on the GPU only (no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("mi355x hot path: tensor must be contiguous")


import os as _os

CHECK_RANGE = _os.environ.get("MD_CHECK_RANGE") == "1"


def check_fp16_range(t: torch.Tensor, what: str) -> None:
    """Debug aid (MD_CHECK_RANGE=1; synchronises the host): the forward products split their operands into fp16 halves, which
    needs |x| < 65504.  Activations and weights of these models are O(1); an UNSCALED input signal is the one
    way to violate it -- the kernels would then return inf where the fp32 reference stays finite."""
    if CHECK_RANGE and t.numel() and float(t.detach().abs().amax()) >= 65504.0:
        raise RuntimeError("mi355x hot path: %s holds |x| >= 65504; the split-fp16 forward products overflow there -- scale the "
                           "input as the reference's data pipeline does (RobustScaler), or use md_set_exact_fp32(1)" % what)


def f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError(f"mi355x hot path: expected float32, got {t.dtype}")
    return t


def make_desc(n, ti, hi, wi, cin, cout, kernel, stride, padding) -> N.MdConvDesc:
    kt, kh, kw = kernel
    st, sh, sw = stride
    pt, ph, pw = padding
    to = (ti + 2 * pt - kt) // st + 1
    ho = (hi + 2 * ph - kh) // sh + 1
    wo = (wi + 2 * pw - kw) // sw + 1
    return N.MdConvDesc(n, ti, hi, wi, cin, to, ho, wo, cout, kt, kh, kw, st, sh, sw, pt, ph, pw)


def view(data: torch.Tensor, scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
         slope: float = 1.0) -> N.MdActView:
    require_cuda(data, scale, shift)
    return N.MdActView(data.data_ptr(), None if scale is None else scale.data_ptr(),
                       None if shift is None else shift.data_ptr(), float(slope))


# ----------------------------------------------------------------------------------------- layout
def to_channels_last(x: torch.Tensor) -> torch.Tensor:
    """(B,C,T,H,W) -> [B,T,H,W,Cp]"""
    require_cuda(x); f32(x)
    B, Cc, T, H, W = x.shape
    out = torch.empty((B, T, H, W, cpad(Cc)), device=x.device, dtype=torch.float32)
    N.check(N.lib().md_nchw_to_cl(_p(x), B, Cc, T * H * W, _p(out), _stream()), "md_nchw_to_cl")
    return out


def from_channels_last(x: torch.Tensor, channels: int) -> torch.Tensor:
    require_cuda(x); f32(x)
    B, T, H, W, Cp = x.shape
    assert Cp == cpad(channels)
    out = torch.empty((B, channels, T, H, W), device=x.device, dtype=torch.float32)
    N.check(N.lib().md_cl_to_nchw(_p(x), B, channels, T * H * W, _p(out), _stream()), "md_cl_to_nchw")
    return out


# ----------------------------------------------------------------------------------------- conv
# Batched packing for the composable models: pack_weights() remembers (descriptor, want_dgrad) per weight tensor; inside
# ``prepacked(module)`` -- the models' forward() -- the operands of every remembered weight are packed up front by ONE batched
# call (md_conv_pack_weights_batch: ~120 tiny launches per SlowFast step become 3) and pack_weights() hands them out.
_pack_stream = [None]   # the stream the packs of the active prepacked() scope were made on
_pack_memo = {}        # id(weight) -> (weak reference to the weight, descriptor fields, want_dgrad)
_pack_ready = None     # inside prepacked(): id(weight) -> (descriptor fields, wf, wd)
_BN_FUSED_FIN_ROWS = int(os.environ.get("MD_BN_FUSED_FIN_ROWS", "0"))   # partial rows up to which bn_backward fuses the finalize
_PREPACK_OFF = os.environ.get("MD_PREPACK") == "0"      # A/B switch: every unit packs its own operands again


def _desc_key(d: N.MdConvDesc):
    return tuple(getattr(d, f) for f, _ in d._fields_)


class prepacked:
    def __init__(self, module: torch.nn.Module):
        ids = getattr(module, "_md_param_ids", None)
        if ids is None:
            ids = frozenset(id(p) for p in module.parameters())
            module.__dict__["_md_param_ids"] = ids
        self.ids = ids

    def __enter__(self):
        global _pack_ready
        self.outer = _pack_ready
        if self.outer is not None or not _pack_memo or _PREPACK_OFF:
            return self
        items = []
        for wid, (ref, key, wd) in list(_pack_memo.items()):
            w = ref()
            if w is None:                              # the parameter is gone (its id may be reused by another object): forget it
                del _pack_memo[wid]
            elif wid in self.ids and w.is_cuda:
                items.append((wid, w, key, wd))
        if not items:
            return self
        L = N.lib()
        n = len(items)
        descs = (N.MdConvDesc * n)()
        wp = (C.c_void_p * n)(); fp = (C.c_void_p * n)(); dp = (C.c_void_p * n)()
        ready = {}
        for i, (wid, w, key, want_d) in enumerate(items):
            for (f, _), v in zip(N.MdConvDesc._fields_, key):
                setattr(descs[i], f, v)
            wc = w.detach().contiguous()
            wf = torch.empty(L.md_conv_wpack_fwd_floats(C.byref(descs[i])), device=w.device, dtype=torch.float32)
            wd = torch.empty(L.md_conv_wpack_dgrad_floats(C.byref(descs[i])), device=w.device, dtype=torch.float32) if want_d else None
            wp[i] = wc.data_ptr(); fp[i] = wf.data_ptr(); dp[i] = wd.data_ptr() if wd is not None else None
            ready[wid] = (key, wf, wd, wc)
        N.check(L.md_conv_pack_weights_batch(n, descs, wp, fp, dp, _stream()), "md_conv_pack_weights_batch")
        _pack_ready = ready
        self.stream = torch.cuda.current_stream(items[0][1].device)
        _pack_stream[0] = self.stream
        return self

    def __exit__(self, *exc):
        global _pack_ready
        _pack_ready = self.outer
        return False


# ----------------------------------------------------------------------------------------- mask-free dropout
# nn.Dropout without a mask tensor (md_dropout_ctr / md_bias_gelu_drop_ctr): the keep / drop decisions are a pure function of
# (a 128-bit key, call-site tag, element index).  ``counter_dropout(device, training)`` is entered by a model's forward: it draws this
# forward's key -- two int64 words from torch's DEVICE generator, one tiny launch per forward -- and numbers the dropout call sites of
# the forward in call order; the backward of THIS forward regenerates the same decisions from the same key tensor whatever runs in
# between.  Because the key comes from torch's generator, ``torch.manual_seed`` governs the stream exactly as it governs nn.Dropout's,
# and a step recorded into a HIP graph draws a new key on every replay (torch advances the generator's offset per replay).
_drop_ctx = None    # inside counter_dropout(): (this forward's key tensor, [next tag])
_CTR_DROPOUT_OFF = os.environ.get("MD_CTR_DROPOUT") == "0"      # A/B switch: masks from torch's generator again


class counter_dropout:
    def __init__(self, device, training: bool):
        self.device = torch.device(device)
        self.on = bool(training) and self.device.type == "cuda" and not _CTR_DROPOUT_OFF

    def __enter__(self):
        global _drop_ctx
        self.outer = _drop_ctx
        if not self.on or self.outer is not None:        # (a nested model shares the enclosing forward's stream of decisions)
            return self
        _drop_ctx = (torch.empty(2, dtype=torch.int64, device=self.device).random_(), [0])
        return self

    def __exit__(self, *exc):
        global _drop_ctx
        _drop_ctx = self.outer
        return False


def dropout_site():
    """(key tensor, tag) of the next dropout call site of the forward in progress, or None outside counter_dropout()."""
    if _drop_ctx is None:
        return None
    tag = _drop_ctx[1][0]
    _drop_ctx[1][0] = tag + 1
    return _drop_ctx[0], tag


class own_packs:
    """Hide the packs of an enclosing ``prepacked`` scope: a branch whose launches are RECORDED (src/utils/graphed.py::GraphedBranch)
    must pack its operands itself, inside the recording -- the buffers of the enclosing scope are temporaries of one eager forward,
    and a graph that baked their addresses would read whatever lives there at replay time."""

    def __enter__(self):
        global _pack_ready
        self.outer = _pack_ready
        _pack_ready = None
        return self

    def __exit__(self, *exc):
        global _pack_ready
        _pack_ready = self.outer
        return False


def pack_weights(d: N.MdConvDesc, w: torch.Tensor, want_dgrad: bool = True, owner: Optional[torch.Tensor] = None):
    """``owner``: the Parameter whose memory ``w`` is a reshaped view of (a Linear's (Dout, Din) weight handed over as a
    (Dout, Din, 1, 1, 1) convolution weight): the batched pre-pack of ``prepacked`` is remembered and looked up under ITS identity."""
    require_cuda(w); f32(w)
    key = _desc_key(d)
    if owner is not None and not (isinstance(owner, torch.nn.Parameter) and owner.is_contiguous() and owner.numel() == w.numel()
                                  and owner.data_ptr() == w.data_ptr()):
        owner = None
    ident = w if owner is None else owner
    if _pack_ready is not None:
        hit = _pack_ready.get(id(ident))
        if hit is not None and hit[0] == key and (hit[2] is not None or not want_dgrad):
            cur = torch.cuda.current_stream(w.device)
            if cur != _pack_stream[0]:                     # a branch on a side stream (src/utils/streams.py) reads packs made on another
                hit[1].record_stream(cur)
                if hit[2] is not None:
                    hit[2].record_stream(cur)
            return hit[1], (hit[2] if want_dgrad else None)
    if isinstance(ident, torch.nn.Parameter):             # (views and temporaries have no stable identity)
        _pack_memo[id(ident)] = (weakref.ref(ident), key, bool(want_dgrad))
    L = N.lib()
    wf = torch.empty(L.md_conv_wpack_fwd_floats(C.byref(d)), device=w.device, dtype=torch.float32)
    wd = torch.empty(L.md_conv_wpack_dgrad_floats(C.byref(d)), device=w.device, dtype=torch.float32) if want_dgrad else None
    N.check(L.md_conv_pack_weights(C.byref(d), _p(w), _p(wf), _p(wd), _stream()), "md_conv_pack_weights")
    return wf, wd


def conv_fwd(d: N.MdConvDesc, x: N.MdActView, wf: torch.Tensor, device, want_stats: bool = True):
    L = N.lib()
    y = torch.empty((d.N, d.To, d.Ho, d.Wo, cpad(d.Cout)), device=device, dtype=torch.float32)
    part = None
    if want_stats:
        part = torch.empty((L.md_conv_fwd_stat_blocks(C.byref(d)), 2, cpad(d.Cout)), device=device, dtype=torch.float32)
    N.check(L.md_conv_fwd(C.byref(d), C.byref(x), _p(wf), _p(y), _p(part), _stream()), "md_conv_fwd")
    return y, part


def conv_dgrad(d: N.MdConvDesc, dy: torch.Tensor, wd: torch.Tensor, out: Optional[torch.Tensor] = None,
               accumulate: bool = False) -> torch.Tensor:
    require_cuda(dy, wd, out)
    if out is None:
        out = torch.empty((d.N, d.Ti, d.Hi, d.Wi, cpad(d.Cin)), device=dy.device, dtype=torch.float32)
        accumulate = False
    N.check(N.lib().md_conv_dgrad(C.byref(d), _p(dy), _p(wd), _p(out), int(accumulate), _stream()), "md_conv_dgrad")
    return out


def conv_dgrad_bnred(d: N.MdConvDesc, dy: torch.Tensor, wd: torch.Tensor, y_view: N.MdActView, st: torch.Tensor,
                     out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """Data gradient with the BatchNorm-backward reduction of the producing unit fused into the epilogue
    (md_conv_dgrad_bnred): returns (g, partial) or None when this geometry has no fused form.  ``y_view`` is the
    producer's raw output with its BatchNorm scale/shift/slope, ``st`` its [mean, invstd, scale, shift] rows."""
    require_cuda(dy, wd, out, st)
    L = N.lib()
    nb = L.md_conv_dgrad_bnred_blocks(C.byref(d))
    if nb <= 0:
        return None
    if out is None:
        out = torch.empty((d.N, d.Ti, d.Hi, d.Wi, cpad(d.Cin)), device=dy.device, dtype=torch.float32)
        accumulate = False
    part = torch.empty((nb, 2, cpad(d.Cin)), device=dy.device, dtype=torch.float32)
    N.check(L.md_conv_dgrad_bnred(C.byref(d), _p(dy), _p(wd), _p(out), int(accumulate), C.byref(y_view), _p(st[0]), _p(st[1]),
                                  _p(part), _stream()), "md_conv_dgrad_bnred")
    return out, part


def bn_backward_from_g(g: torch.Tensor, part: torch.Tensor, main: N.MdActView, st: torch.Tensor, Cc: int):
    """Finalize + apply after a fused reduction: returns (d_raw, dgamma, dbeta)."""
    require_cuda(g, part, st)
    L = N.lib()
    rows = g.numel() // g.shape[-1]
    Cp = cpad(Cc)
    dgamma = torch.empty(Cc, device=g.device, dtype=torch.float32)
    dbeta = torch.empty(Cc, device=g.device, dtype=torch.float32)
    coef = torch.empty((2, Cp), device=g.device, dtype=torch.float32)
    N.check(L.md_bn_bwd_finalize(_p(part), part.shape[0], Cc, rows, _p(dgamma), _p(dbeta), _p(coef), _stream()), "md_bn_bwd_finalize")
    d_raw = torch.empty_like(g)
    N.check(L.md_bn_bwd_apply_g(_p(g), C.byref(main), _p(st[0]), _p(st[1]), _p(coef), rows, Cc, _p(d_raw), _stream()),
            "md_bn_bwd_apply_g")
    return d_raw, dgamma, dbeta


def bn_apply_fmt(dA: torch.Tensor, main: N.MdActView, st: torch.Tensor, coef: torch.Tensor, Cc: int, split_out: bool,
                 g_in: bool = False) -> torch.Tensor:
    """The apply pass of BatchNorm-backward alone (md_bn_bwd_apply_fmt): d_raw in fp32 or in the pre-split bf16 format
    (same bytes; returned as a float32-typed tensor either way)."""
    require_cuda(dA, st, coef)
    rows = dA.numel() // dA.shape[-1]
    out = torch.empty_like(dA)
    N.check(N.lib().md_bn_bwd_apply_fmt(_p(dA), int(g_in), C.byref(main), None, 1.0, _p(st[0]), _p(st[1]), _p(coef), rows, Cc,
                                        _p(out), int(split_out), None, _stream()), "md_bn_bwd_apply_fmt")
    return out


def conv_dgrad_fmt(d: N.MdConvDesc, dy: torch.Tensor, dy_split: bool, wd: torch.Tensor, out: Optional[torch.Tensor] = None,
                   accumulate: bool = False) -> torch.Tensor:
    require_cuda(dy, wd, out)
    if out is None:
        out = torch.empty((d.N, d.Ti, d.Hi, d.Wi, cpad(d.Cin)), device=dy.device, dtype=torch.float32)
        accumulate = False
    N.check(N.lib().md_conv_dgrad_fmt(C.byref(d), _p(dy), int(dy_split), _p(wd), _p(out), int(accumulate), None, None, None, None,
                                      _stream()), "md_conv_dgrad_fmt")
    return out


def conv_wgrad_fmt(d: N.MdConvDesc, x: N.MdActView, dy: torch.Tensor, dy_split: bool) -> torch.Tensor:
    require_cuda(dy)
    dw = torch.empty((d.Cout, d.Cin, d.kt, d.kh, d.kw), device=dy.device, dtype=torch.float32)
    nws = N.lib().md_conv_wgrad_workspace_floats(C.byref(d))
    ws = torch.empty(nws, device=dy.device, dtype=torch.float32) if nws else None
    N.check(N.lib().md_conv_wgrad_fmt(C.byref(d), C.byref(x), _p(dy), int(dy_split), _p(dw), _p(ws), _stream()), "md_conv_wgrad_fmt")
    return dw


def bn_act_split(v: N.MdActView, rows: int, Cc: int, device) -> torch.Tensor:
    """leaky(scale * y + shift) (or the tensor as it is, scale None) as the pre-split bf16 copy the weight-gradient kernels stage
    by plain copy (md_bn_act_split); returned as a float32 tensor of the same byte size."""
    out = torch.empty(N.lib().md_bn_act_split_floats(rows, Cc), device=device, dtype=torch.float32)
    N.check(N.lib().md_bn_act_split(C.byref(v), rows, Cc, _p(out), _stream()), "md_bn_act_split")
    return out


def conv_wgrad_xsplit(d: N.MdConvDesc, x_split: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """Weight gradient reading the pre-split copy of its input (md_conv_wgrad_fmt2, x_split = 1)."""
    require_cuda(dy, x_split)
    if not N.lib().md_conv_wgrad_xsplit_ok(C.byref(d)):
        raise RuntimeError("mi355x hot path: this geometry has no pre-split weight-gradient form")
    dw = torch.empty((d.Cout, d.Cin, d.kt, d.kh, d.kw), device=dy.device, dtype=torch.float32)
    nws = N.lib().md_conv_wgrad_workspace_floats(C.byref(d))
    ws = torch.empty(nws, device=dy.device, dtype=torch.float32) if nws else None
    xv = view(x_split)
    N.check(N.lib().md_conv_wgrad_fmt2(C.byref(d), C.byref(xv), 1, _p(dy), 0, _p(dw), _p(ws), _stream()), "md_conv_wgrad_fmt2")
    return dw


def conv_wgrad(d: N.MdConvDesc, x: N.MdActView, dy: torch.Tensor) -> torch.Tensor:
    require_cuda(dy)
    dw = torch.empty((d.Cout, d.Cin, d.kt, d.kh, d.kw), device=dy.device, dtype=torch.float32)
    nws = N.lib().md_conv_wgrad_workspace_floats(C.byref(d))
    ws = torch.empty(nws, device=dy.device, dtype=torch.float32) if nws else None
    N.check(N.lib().md_conv_wgrad(C.byref(d), C.byref(x), _p(dy), _p(dw), _p(ws), _stream()), "md_conv_wgrad")
    return dw


# ----------------------------------------------------------------------------------------- batch norm
def bn_finalize(part: torch.Tensor, Cc: int, count: int, gamma, beta, rmean=None, rvar=None, eps=1e-5, momentum=0.1):
    require_cuda(part, gamma, beta, rmean, rvar)
    Cp = cpad(Cc)
    st = torch.empty((4, Cp), device=part.device, dtype=torch.float32)   # mean, invstd, scale, shift
    N.check(N.lib().md_bn_finalize(_p(part), part.shape[0], Cc, count, _p(gamma), _p(beta), eps, momentum, _p(rmean),
                                   _p(rvar), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), _stream()), "md_bn_finalize")
    return st


def bn_act(v: N.MdActView, like: torch.Tensor, Cc: int) -> torch.Tensor:
    out = torch.empty_like(like)
    rows = like.numel() // like.shape[-1]
    N.check(N.lib().md_bn_act(C.byref(v), rows, Cc, _p(out), _stream()), "md_bn_act")
    return out


def residual_fwd(skip: N.MdActView, main: N.MdActView, alpha: float, like: torch.Tensor, Cc: int) -> torch.Tensor:
    out = torch.empty_like(like)
    rows = like.numel() // like.shape[-1]
    N.check(N.lib().md_residual_fwd(C.byref(skip), C.byref(main), float(alpha), rows, Cc, _p(out), _stream()),
            "md_residual_fwd")
    return out


def bn_backward(dA: torch.Tensor, main: N.MdActView, st: torch.Tensor, Cc: int, skip: Optional[N.MdActView] = None,
                alpha: float = 1.0, fused_finalize: Optional[bool] = None):
    """Returns (d_raw, dS or None, dgamma, dbeta).  ``fused_finalize`` None: decided by the number of partial rows (see below)."""
    require_cuda(dA, st)
    L = N.lib()
    rows = dA.numel() // dA.shape[-1]
    Cp = cpad(Cc)
    nb = L.md_bn_bwd_blocks(rows, Cc)
    part = torch.empty((nb, 2, Cp), device=dA.device, dtype=torch.float32)
    sk = C.byref(skip) if skip is not None else None
    N.check(L.md_bn_bwd_reduce(_p(dA), C.byref(main), sk, float(alpha), _p(st[0]), _p(st[1]), rows, Cc, _p(part), _stream()),
            "md_bn_bwd_reduce")
    dgamma = torch.empty(Cc, device=dA.device, dtype=torch.float32)
    dbeta = torch.empty(Cc, device=dA.device, dtype=torch.float32)
    d_raw = torch.empty_like(dA)
    dS = torch.empty_like(dA) if skip is not None else None
    if fused_finalize is None:
        fused_finalize = nb <= _BN_FUSED_FIN_ROWS
    if fused_finalize:
        # the apply pass sums the partial rows itself: one launch instead of finalize + apply.  Every workgroup of the apply pass
        # repeats the sum, so it pays only where the partial buffer is short (the small tensors of the composable models: one
        # dependent launch less per unit); measured slower on the R(2+1)D step's big tensors (profiles/r03_bn_fused_finalize.txt)
        N.check(L.md_bn_bwd_apply_fused(_p(dA), 0, C.byref(main), sk, float(alpha), _p(st[0]), _p(st[1]), _p(part), nb, rows,
                                        _p(dgamma), _p(dbeta), rows, Cc, _p(d_raw), _p(dS), _stream()), "md_bn_bwd_apply_fused")
        return d_raw, dS, dgamma, dbeta
    coef = torch.empty((2, Cp), device=dA.device, dtype=torch.float32)
    N.check(L.md_bn_bwd_finalize(_p(part), nb, Cc, rows, _p(dgamma), _p(dbeta), _p(coef), _stream()), "md_bn_bwd_finalize")
    N.check(L.md_bn_bwd_apply(_p(dA), C.byref(main), sk, float(alpha), _p(st[0]), _p(st[1]), _p(coef), rows, Cc, _p(d_raw),
                              _p(dS), _stream()), "md_bn_bwd_apply")
    return d_raw, dS, dgamma, dbeta


# ----------------------------------------------------------------------------------------- pool / head / loss
def avgpool_fwd(x: torch.Tensor, Cc: int) -> torch.Tensor:
    require_cuda(x)
    B = x.shape[0]
    thw = x.shape[1] * x.shape[2] * x.shape[3]
    feat = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
    N.check(N.lib().md_avgpool_fwd(_p(x), B, Cc, thw, _p(feat), _stream()), "md_avgpool_fwd")
    return feat


def avgpool_bwd(dfeat: torch.Tensor, shape: Sequence[int]) -> torch.Tensor:
    require_cuda(dfeat)
    B, T, H, W, Cp = shape
    dx = torch.empty(tuple(shape), device=dfeat.device, dtype=torch.float32)
    N.check(N.lib().md_avgpool_bwd(_p(dfeat), B, dfeat.shape[1], T * H * W, _p(dx), _stream()), "md_avgpool_bwd")
    return dx


KIND = {"focal": 0, "ldam": 1, "ce": 2}


def softmax_loss(kind: str, logits: torch.Tensor, target: torch.Tensor, weight: Optional[torch.Tensor],
                 margins: Optional[torch.Tensor], gamma_or_s: float, want_grad: bool = True):
    """Returns (loss[1], dlogits or None, pred int64[B])."""
    require_cuda(logits, target, weight, margins); f32(logits)
    if target.dtype != torch.int64:
        raise RuntimeError("target must be int64")
    B, K = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    dl = torch.empty_like(logits) if want_grad else None
    pred = torch.empty(B, device=logits.device, dtype=torch.int64)
    N.check(N.lib().md_softmax_loss(KIND[kind], _p(logits), _p(target), B, K, _p(weight), _p(margins), float(gamma_or_s),
                                    _p(loss), _p(dl), _p(pred), _stream()), "md_softmax_loss")
    return loss, dl, pred

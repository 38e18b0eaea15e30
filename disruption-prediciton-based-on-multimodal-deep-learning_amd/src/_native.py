"""ctypes binding of include/mi355x_disrupt.h (the C-ABI boundary of the hot path).

Loading is lazy and LOUD: if the shared library is missing or lacks a symbol the first call raises
``RuntimeError`` -- the product path never falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

# PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE this library is
# dlopen'ed so that both resolve to the same runtime; loaded the other way round the process holds two HIP runtimes
# and every launch on a torch stream fails (measured: "kernel launch failed" from md_plan_forward).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libmi355x_disrupt.so")

c_float_p = C.POINTER(C.c_float)
c_void_p = C.c_void_p


class MdConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("N", "Ti", "Hi", "Wi", "Cin", "To", "Ho", "Wo", "Cout", "kt", "kh", "kw", "st", "sh", "sw", "pt", "ph", "pw")]


class MdActView(C.Structure):
    _fields_ = [("data", c_void_p), ("scale", c_void_p), ("shift", c_void_p), ("slope", C.c_float)]


class MdTrainStepArgs(C.Structure):
    """include/mi355x_disrupt.h::MdTrainStepArgs, field for field."""
    _fields_ = ([(n, C.c_int32) for n in ("B", "Hd", "K", "loss_kind")] +
                [(n, c_void_p) for n in ("x", "target", "w", "gamma", "beta", "rmean", "rvar", "dw", "dgamma", "dbeta",
                                         "w0", "b0", "hgamma", "hbeta", "w1", "b1", "hrmean", "hrvar",
                                         "dw0", "db0", "dhgamma", "dhbeta", "dw1", "db1")] +
                [(n, C.c_float) for n in ("head_alpha", "head_eps", "head_momentum", "gamma_or_s")] +
                [(n, c_void_p) for n in ("class_weight", "margins", "feat", "dfeat", "logits", "dlogits", "head_save", "loss", "pred",
                                         "workspace", "counters")] +
                [("ncounters", C.c_int32), ("opt_nchunks", C.c_int32)] +
                [(n, c_void_p) for n in ("opt_tensors", "opt_chunks", "opt_partial")] +
                [(n, C.c_float) for n in ("max_norm", "lr", "beta1", "beta2", "eps", "weight_decay")] +
                [("opt_step", C.c_int64), ("ok_flag", c_void_p)])


# name -> (restype, argtypes); mirrors include/mi355x_disrupt.h one to one
_P = c_void_p
_I32, _I64, _F, _SZ = C.c_int32, C.c_int64, C.c_float, C.c_size_t
_DESC, _VIEW = C.POINTER(MdConvDesc), C.POINTER(MdActView)
SIGNATURES = {
    "md_version": (C.c_int, [C.POINTER(C.c_char_p)]),
    "md_set_exact_fp32": (C.c_int, [C.c_int]),
    "md_get_exact_fp32": (C.c_int, []),
    "md_set_pers_grid": (C.c_int, [C.c_int]),
    "md_set_wgrad_form": (C.c_int, [C.c_int]),
    "md_conv_wpack_fwd_floats": (_SZ, [_DESC]),
    "md_conv_wpack_dgrad_floats": (_SZ, [_DESC]),
    "md_conv_pack_weights": (C.c_int, [_DESC, _P, _P, _P, _P]),
    "md_conv_pack_weights_batch": (C.c_int, [_I32, _P, _P, _P, _P, _P]),
    "md_conv_fwd_stat_blocks": (_I32, [_DESC]),
    "md_conv_fwd": (C.c_int, [_DESC, _VIEW, _P, _P, _P, _P]),
    "md_conv_dgrad": (C.c_int, [_DESC, _P, _P, _P, C.c_int, _P]),
    "md_conv_wgrad_workspace_floats": (_SZ, [_DESC]),
    "md_conv_wgrad": (C.c_int, [_DESC, _VIEW, _P, _P, _P, _P]),
    "md_bn_finalize": (C.c_int, [_P, _I32, _I32, _I64, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    "md_bn_eval_params": (C.c_int, [_I32, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P]),
    "md_bn_act": (C.c_int, [_VIEW, _I64, _I32, _P, _P]),
    "md_residual_fwd": (C.c_int, [_VIEW, _VIEW, _F, _I64, _I32, _P, _P]),
    "md_bn_bwd_blocks": (_I32, [_I64, _I32]),
    "md_bn_bwd_reduce": (C.c_int, [_P, _VIEW, _VIEW, _F, _P, _P, _I64, _I32, _P, _P]),
    "md_bn_bwd_finalize": (C.c_int, [_P, _I32, _I32, _I64, _P, _P, _P, _P]),
    "md_bn_bwd_apply": (C.c_int, [_P, _VIEW, _VIEW, _F, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "md_conv_dgrad_bnred_blocks": (_I32, [_DESC]),
    "md_conv_dgrad_bnred": (C.c_int, [_DESC, _P, _P, _P, C.c_int, _VIEW, _P, _P, _P, _P]),
    "md_bn_bwd_apply_g": (C.c_int, [_P, _VIEW, _P, _P, _P, _I64, _I32, _P, _P]),
    "md_conv_split_dy_ok": (C.c_int, [_DESC, C.c_int]),
    "md_bn_bwd_apply_fmt": (C.c_int, [_P, C.c_int, _VIEW, _VIEW, _F, _P, _P, _P, _I64, _I32, _P, C.c_int, _P, _P]),
    "md_bn_bwd_apply_fused": (C.c_int, [_P, C.c_int, _VIEW, _VIEW, _F, _P, _P, _P, _I32, _I64, _P, _P, _I64, _I32, _P, _P, _P]),
    "md_conv_dgrad_fmt": (C.c_int, [_DESC, _P, C.c_int, _P, _P, C.c_int, _VIEW, _P, _P, _P, _P]),
    "md_conv_wgrad_fmt": (C.c_int, [_DESC, _VIEW, _P, C.c_int, _P, _P, _P]),
    "md_conv_wgrad_fmt2": (C.c_int, [_DESC, _VIEW, C.c_int, _P, C.c_int, _P, _P, _P]),
    "md_conv_wgrad_xsplit_ok": (C.c_int, [_DESC]),
    "md_bn_act_split_floats": (_SZ, [_I64, _I32]),
    "md_bn_act_split": (C.c_int, [_VIEW, _I64, _I32, _P, _P]),
    "md_nchw_to_cl": (C.c_int, [_P, _I32, _I32, _I64, _P, _P]),
    "md_cl_to_nchw": (C.c_int, [_P, _I32, _I32, _I64, _P, _P]),
    "md_cat_cl": (C.c_int, [_P, _I32, _P, _I32, _I64, _P, _P]),
    "md_split_cl": (C.c_int, [_P, _I32, _I32, _I64, _P, _P, _P]),
    "md_avgpool_fwd": (C.c_int, [_P, _I32, _I32, _I64, _P, _P]),
    "md_avgpool_bwd": (C.c_int, [_P, _I32, _I32, _I64, _P, _P]),
    "md_head_save_floats": (_SZ, [_I32, _I32, _I32]),
    "md_head_fwd": (C.c_int, [_P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _F, _F, _F, C.c_int, _P, _P, _P, _P, _P]),
    "md_head_bwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "md_softmax_loss": (C.c_int, [_I32, _P, _P, _I32, _I32, _P, _P, _F, _P, _P, _P, _P]),
    "md_plan_create": (C.c_int, [_I32, _I32, _I32, _I32, C.POINTER(_I32), _F, C.POINTER(_P)]),
    "md_plan_destroy": (None, [_P]),
    "md_plan_num_units": (_I32, [_P]),
    "md_plan_unit_desc": (C.c_int, [_P, _I32, _DESC]),
    "md_plan_workspace_bytes": (_SZ, [_P]),
    "md_plan_unit_layout": (C.c_int, [_P, _I32, C.POINTER(_SZ), C.POINTER(_SZ), C.POINTER(_I64), C.POINTER(_I32)]),
    "md_plan_num_z": (_I32, [_P]),
    "md_plan_z_layout": (C.c_int, [_P, _I32, C.POINTER(_SZ), C.POINTER(_I64), C.POINTER(_I32)]),
    "md_plan_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P]),
    "md_plan_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "md_plan_backward_range": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _P]),
    "md_plan_feat_dim": (_I32, [_P]),
    "md_plan_train_step": (C.c_int, [_P, C.POINTER(MdTrainStepArgs), _P]),
    "md_plan_profile_enable": (C.c_int, [_P, C.c_int]),
    "md_plan_profile_reserve": (C.c_int, [_P, _I32]),
    "md_plan_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "md_plan_use_side_stream": (C.c_int, [_P, _I32]),
    "md_plan_defer_join": (C.c_int, [_P, _I32]),
    "md_plan_side_stream": (C.c_void_p, [_P]),
    "md_plan_join": (C.c_int, [_P, _P]),
    "md_swish_fwd": (C.c_int, [_P, C.c_int64, _P, _P]),
    "md_swish_bwd": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "md_add_noise": (C.c_int, [_P, _P, _F, _F, C.c_int64, _P, _P]),
    "md_se_swish_fwd": (C.c_int, [_P, _I32, _I32, C.c_int64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "md_se_swish_bwd": (C.c_int, [_P, _P, _I32, _I32, C.c_int64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "md_se_scale_fwd": (C.c_int, [_P, _I32, _I32, C.c_int64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "md_se_scale_bwd": (C.c_int, [_P, _P, _I32, _I32, C.c_int64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "md_mask_scale": (C.c_int, [_P, _P, _F, C.c_int64, _P, _P]),
    "md_add_relu_fwd": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "md_add_leaky_fwd": (C.c_int, [_P, _P, _F, _I64, _P, _P]),
    "md_add_leaky_bwd": (C.c_int, [_P, _P, _F, _I64, _P, _P]),
    "md_add_relu_bwd": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "md_maxpool_1x3x3_fwd": (C.c_int, [_P, C.c_int64, _I32, _I32, _P, _P, _P]),
    "md_maxpool_1x3x3_bwd": (C.c_int, [_P, _P, C.c_int64, _I32, _I32, _P, _P]),
    "md_rowmean_fwd": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "md_rowmean_bwd": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    "md_patch_embed_fwd": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I32, _P, _P, _P, _P, _I32, _P, _P]),
    "md_patch_embed_wgrad_workspace_floats": (_SZ, [_I32, _I32, _I32, _I32, _I32, _I32, _I32]),
    "md_patch_embed_wgrad": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _I32, _P, _I32, _P, _P, _P]),
    "md_channel_bias_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "md_channel_bias_bwd": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _P]),
    "md_channel_bias_bwd_scratch_floats": (_SZ, [_I32, _I32, _I32]),
    "md_seq_sum_fwd": (C.c_int, [_P, _I32, _I32, _I32, _F, _P, _P]),
    "md_seq_sum_bwd": (C.c_int, [_P, _I32, _I32, _I32, _F, _P, _P]),
    "md_add_layernorm_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int64, _I32, _F, _P, _P, _P, _P, _P]),
    "md_add_layernorm_bwd_scratch_floats": (_SZ, [C.c_int64, _I32]),
    "md_add_layernorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _I32, _P, _P, _P, _P, _P]),
    "md_attention_fwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "md_attention_bwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "md_attention_lse_supported": (C.c_int32, [_I32, _I32, _I32]),
    "md_attention_lse_fwd": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "md_attention_lse_bwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "md_elu": (C.c_int, [_P, _P, _F, C.c_int64, _P, _P]),
    "md_lstm_rec_supported": (C.c_int, [_I32]),
    "md_lstm_rec_fwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "md_lstm_rec_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "md_lstm_rec_fwd2": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _P]),
    "md_lstm_rec_bwd2": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _P]),
    "md_clip_preprocess": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, C.POINTER(C.c_float), _I32, _P, _P]),
    "md_clip_augment_preprocess": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, C.POINTER(C.c_float), _I32, _P, _P, _P, _P]),
    "md_multinomial_shard_count": (C.c_int64, [C.c_int64, _I32, _I32]),
    "md_multinomial_shard": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _I32, _I32, _P, _P, _P]),
    "md_outer_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "md_outer_bwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "md_gelu": (C.c_int, [_P, _P, _I32, C.c_int64, _P, _P]),
    "md_bias_gelu_drop": (C.c_int, [_P, _P, _P, _P, C.c_float, _I32, _I64, _I32, _P, _P]),
    "md_dropout_ctr": (C.c_int, [_P, _P, _I32, _F, _F, _I64, _P, _P]),
    "md_branch_layernorm_supported": (C.c_int, [_I64, _I32]),
    "md_branch_layernorm_fwd": (C.c_int, [_P, _P, _P, _I32, _F, _P, _P, _P, _I64, _I32, _F, _P, _P, _P, _P, _P]),
    "md_branch_layernorm_bwd_scratch_floats": (_SZ, [_I64, _I32]),
    "md_branch_layernorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _F, _I64, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "md_bias_gelu_drop_ctr": (C.c_int, [_P, _P, _P, _I32, _F, _P, _F, _I32, _I64, _I32, _P, _P]),
    "md_lstm_fwd": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "md_lstm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P]),
    "md_opt_chunk_elems": (C.c_int, []),
    "md_opt_grad_norm": (C.c_int, [_P, _P, _I32, _F, _P, _P, _P]),
    "md_opt_adamw_step": (C.c_int, [_P, _P, _I32, _P, _F, _F, _F, _F, _F, C.c_int64, _P]),
    "md_opt_adamw_step_if": (C.c_int, [_P, _P, _I32, _P, _F, _F, _F, _F, _F, C.c_int64, _P, _P]),
}

ERRORS = {-1: "bad shape", -2: "unsupported", -3: "workspace", -4: "kernel launch failed", -5: "null pointer"}

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """The loaded library with every prototype declared; raises RuntimeError if it cannot be loaded."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"MI355X HIP library not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (or `make -C csrc`). There is no CPU fallback for this path.")
                try:
                    L = C.CDLL(LIB_PATH)
                except OSError as e:  # pragma: no cover
                    raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
                for name, (res, args) in SIGNATURES.items():
                    try:
                        f = getattr(L, name)
                    except AttributeError as e:
                        raise RuntimeError(f"{LIB_PATH} does not export {name}") from e
                    f.restype = res
                    f.argtypes = args
                _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}")

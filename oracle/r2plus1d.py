"""Oracle: functional fp32 CPU restatement of the reference R(2+1)D classifier.

Test infrastructure (see ``oracle/__init__.py``).  The network is expressed as a
flat list of convolution "units" driven by the reference's state-dict key names,
so the same description can be checked against a reference ``state_dict``.

Reference anchors (relative to the reference root):
  * Conv3dBlock            src/models/R2Plus1D.py:25-58   conv(bias=False) -> BatchNorm3d -> LeakyReLU
  * SpatioTemporalConv     src/models/R2Plus1D.py:115-162 (1,k,k) block then (k,1,1) block, mid-channel formula :150-155
  * SpatioTemporalResBlock src/models/R2Plus1D.py:164-187 conv1 -> conv2 (+ 1x1x1 strided skip) -> add -> LeakyReLU(alpha)
  * SpatioTemporalResLayer src/models/R2Plus1D.py:190-204
  * R2Plus1DNet            src/models/R2Plus1D.py:207-226 stem (45 mid channels, 7x7/s2) + 4 stages + global average pool
  * R2Plus1DClassifier     src/models/R2Plus1D.py:228-283 head Linear -> BatchNorm1d -> ELU(alpha) -> Linear

Quirk restated on purpose: the residual blocks build their SpatioTemporalConv
children WITHOUT passing ``alpha`` (R2Plus1D.py:172-178), so every conv unit
inside a residual block uses LeakyReLU(0.01); only the stem units and the
block-closing activation use the constructor's ``alpha``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
INNER_SLOPE = 0.01  # default alpha of SpatioTemporalConv (R2Plus1D.py:116)


@dataclass
class ConvUnit:
    """One Conv3dBlock: conv + BN + LeakyReLU."""
    name: str            # state-dict prefix, e.g. "res2plus1d.conv1.spatio_conv"
    cin: int
    cout: int
    kernel: Tuple[int, int, int]
    stride: Tuple[int, int, int]
    padding: Tuple[int, int, int]
    slope: float


def _mid_channels(k: Tuple[int, int, int], cin: int, cout: int) -> int:
    # R2Plus1D.py:150-155
    return int(math.floor((k[0] * k[1] * k[2] * cin * cout) / (k[1] * k[2] * cin + k[0] * cout)))


def st_conv_units(prefix: str, cin: int, cout: int, k: int, stride: int, pad: int) -> List[ConvUnit]:
    """Non-stem SpatioTemporalConv (R2Plus1D.py:138-157): spatial unit then temporal unit."""
    kk = (k, k, k)
    mid = _mid_channels(kk, cin, cout)
    return [
        ConvUnit(prefix + ".spatio_conv", cin, mid, (1, k, k), (1, stride, stride), (0, pad, pad), INNER_SLOPE),
        ConvUnit(prefix + ".temporal_conv", mid, cout, (k, 1, 1), (stride, 1, 1), (pad, 0, 0), INNER_SLOPE),
    ]


def stem_units(alpha: float) -> List[ConvUnit]:
    """is_first SpatioTemporalConv (R2Plus1D.py:125-137, :210)."""
    p = "res2plus1d.conv1"
    return [
        ConvUnit(p + ".spatio_conv", 3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3), alpha),
        ConvUnit(p + ".temporal_conv", 45, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), alpha),
    ]


STAGES = [("conv2", 32, 32, False), ("conv3", 32, 64, True), ("conv4", 64, 64, True), ("conv5", 64, 128, True)]


def _bn(y: torch.Tensor, sd: Dict[str, torch.Tensor], bufs: Dict[str, torch.Tensor], name: str, training: bool):
    return F.batch_norm(
        y, bufs[name + ".running_mean"], bufs[name + ".running_var"],
        sd[name + ".weight"], sd[name + ".bias"], training, BN_MOMENTUM, BN_EPS)


def _leaky(pre, slope: float, name: str, tap, force):
    """LeakyReLU with two diagnostic hooks (tests/kink_util.py).  ``tap[name]`` receives the pre-activation; ``force[name]``
    (bool, True = positive branch) replaces the sign test, i.e. evaluates the SAME piecewise-linear function on a given
    activation pattern -- used to check that two evaluations differ only by which side of zero a few near-zero
    pre-activations fell on (LeakyReLU's derivative jumps by 1/slope there)."""
    if tap is not None:
        tap[name] = pre.detach()
    if force is not None and name in force:
        return torch.where(force[name], pre, pre * slope)
    return F.leaky_relu(pre, slope)


def run_unit(x, u: ConvUnit, sd, bufs, training: bool, tap=None, force=None):
    y = F.conv3d(x, sd[u.name + ".conv.weight"], None, u.stride, u.padding)
    y = _bn(y, sd, bufs, u.name + ".bn", training)
    if training and (u.name + ".bn.num_batches_tracked") in bufs:
        bufs[u.name + ".bn.num_batches_tracked"] += 1
    return _leaky(y, u.slope, u.name, tap, force)


def res_block(x, prefix: str, cin: int, cout: int, downsample: bool, alpha: float, sd, bufs, training: bool,
              tap=None, force=None):
    """SpatioTemporalResBlock.forward (R2Plus1D.py:181-187)."""
    s = 2 if downsample else 1
    r = x
    for u in st_conv_units(prefix + ".conv1", cin, cout, 3, s, 1):
        r = run_unit(r, u, sd, bufs, training, tap, force)
    for u in st_conv_units(prefix + ".conv2", cout, cout, 3, 1, 1):
        r = run_unit(r, u, sd, bufs, training, tap, force)
    if downsample:
        for u in st_conv_units(prefix + ".downsample_conv", cin, cout, 1, 2, 0):
            x = run_unit(x, u, sd, bufs, training, tap, force)
    return _leaky(x + r, alpha, prefix + ".relu", tap, force)


def block_prefixes(layer_sizes) -> List[str]:
    """Residual blocks in execution order (their closing activation is named ``<prefix>.relu`` in tap / force)."""
    out = []
    for (stage, _, _, _), n in zip(STAGES, layer_sizes):
        p = "res2plus1d." + stage
        out += [p + ".block1"] + [f"{p}.blocks.{i}" for i in range(n - 1)]
    return out


def trunk_forward(x, sd, bufs, layer_sizes, alpha: float, training: bool, tap=None, force=None):
    """R2Plus1DNet.forward (R2Plus1D.py:217-226).  x: (B,3,T,H,W) fp32 -> (B,128)."""
    for u in stem_units(alpha):
        x = run_unit(x, u, sd, bufs, training, tap, force)
    for (stage, cin, cout, down), n in zip(STAGES, layer_sizes):
        p = "res2plus1d." + stage
        x = res_block(x, p + ".block1", cin, cout, down, alpha, sd, bufs, training, tap, force)
        for i in range(n - 1):
            x = res_block(x, f"{p}.blocks.{i}", cout, cout, False, alpha, sd, bufs, training, tap, force)
    return x.mean(dim=(2, 3, 4))  # AdaptiveAvgPool3d(1) + view  (R2Plus1D.py:215,224-225)


def head_forward(f, sd, bufs, alpha: float, training: bool):
    """R2Plus1DClassifier.linear (R2Plus1D.py:243-248)."""
    h = F.linear(f, sd["linear.0.weight"], sd["linear.0.bias"])
    h = F.batch_norm(h, bufs["linear.1.running_mean"], bufs["linear.1.running_var"],
                     sd["linear.1.weight"], sd["linear.1.bias"], training, BN_MOMENTUM, BN_EPS)
    if training and "linear.1.num_batches_tracked" in bufs:
        bufs["linear.1.num_batches_tracked"] += 1
    h = F.elu(h, alpha)
    return F.linear(h, sd["linear.3.weight"], sd["linear.3.bias"])


def classifier_forward(x, sd, bufs, layer_sizes, alpha: float, training: bool = True, tap=None, force=None):
    """R2Plus1DClassifier.forward (R2Plus1D.py:280-283)."""
    return head_forward(trunk_forward(x, sd, bufs, layer_sizes, alpha, training, tap, force), sd, bufs, alpha, training)


def all_units(layer_sizes, alpha: float) -> List[ConvUnit]:
    """Every conv unit in execution order (skip path listed after the main path of its block)."""
    out = list(stem_units(alpha))
    for (stage, cin, cout, down), n in zip(STAGES, layer_sizes):
        p = "res2plus1d." + stage
        blocks = [(p + ".block1", cin, cout, down)] + [(f"{p}.blocks.{i}", cout, cout, False) for i in range(n - 1)]
        for bp, bi, bo, bd in blocks:
            out += st_conv_units(bp + ".conv1", bi, bo, 3, 2 if bd else 1, 1)
            out += st_conv_units(bp + ".conv2", bo, bo, 3, 1, 1)
            if bd:
                out += st_conv_units(bp + ".downsample_conv", bi, bo, 1, 2, 0)
    return out


def param_shapes(layer_sizes, alpha: float = 0.01, num_classes: int = 2) -> Dict[str, Tuple[int, ...]]:
    """Shapes of every learnable parameter, keyed like the reference state_dict."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    for u in all_units(layer_sizes, alpha):
        shapes[u.name + ".conv.weight"] = (u.cout, u.cin) + u.kernel
        shapes[u.name + ".bn.weight"] = (u.cout,)
        shapes[u.name + ".bn.bias"] = (u.cout,)
    shapes["linear.0.weight"] = (64, 128)
    shapes["linear.0.bias"] = (64,)
    shapes["linear.1.weight"] = (64,)
    shapes["linear.1.bias"] = (64,)
    shapes["linear.3.weight"] = (num_classes, 64)
    shapes["linear.3.bias"] = (num_classes,)
    return shapes


def buffer_shapes(layer_sizes, alpha: float = 0.01) -> Dict[str, Tuple[int, ...]]:
    shapes: Dict[str, Tuple[int, ...]] = {}
    for u in all_units(layer_sizes, alpha):
        shapes[u.name + ".bn.running_mean"] = (u.cout,)
        shapes[u.name + ".bn.running_var"] = (u.cout,)
        shapes[u.name + ".bn.num_batches_tracked"] = ()
    shapes["linear.1.running_mean"] = (64,)
    shapes["linear.1.running_var"] = (64,)
    shapes["linear.1.num_batches_tracked"] = ()
    return shapes


def synth_state(layer_sizes, seed: int, alpha: float = 0.01):
    """Deterministic synthetic parameters (NumPy PCG64 stream, independent of torch's RNG).

    Convolutions ~ N(0, 2/fan_out-ish) like Kaiming-normal (R2Plus1D.py:267-273 uses
    kaiming_normal_ on torch's RNG; the exact draw is irrelevant for parity, only that the
    reference and the build load the SAME numbers).  BN gamma/beta are perturbed away from
    1/0 so the affine path is exercised.
    """
    import numpy as np
    rng = np.random.default_rng(seed)
    params: Dict[str, torch.Tensor] = {}
    for k, shp in param_shapes(layer_sizes, alpha).items():
        if k.endswith("conv.weight"):
            fan_in = shp[1] * shp[2] * shp[3] * shp[4]
            a = rng.standard_normal(shp) * math.sqrt(2.0 / fan_in)
        elif k.endswith("bn.weight") or k == "linear.1.weight":
            a = 1.0 + 0.2 * rng.standard_normal(shp)
        elif k.endswith("bn.bias") or k == "linear.1.bias":
            a = 0.1 * rng.standard_normal(shp)
        elif k.endswith(".weight"):
            a = rng.standard_normal(shp) / math.sqrt(shp[1])
        else:
            a = 0.05 * rng.standard_normal(shp)
        params[k] = torch.from_numpy(a.astype("float32"))
    bufs: Dict[str, torch.Tensor] = {}
    for k, shp in buffer_shapes(layer_sizes, alpha).items():
        if k.endswith("running_mean"):
            bufs[k] = torch.zeros(shp)
        elif k.endswith("running_var"):
            bufs[k] = torch.ones(shp)
        else:
            bufs[k] = torch.zeros((), dtype=torch.int64)
    return params, bufs


def synth_clip(B: int, T: int, S: int, seed: int) -> torch.Tensor:
    """Synthetic IVIS-like clip batch: uniform-int[0,255] minus BGR means, (B,3,T,S,S) fp32.

    Mirrors what DatasetForVideo hands the model (src/dataset.py:105-110,201-205,229-230).
    """
    import numpy as np
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, size=(B, 3, T, S, S)).astype("float32")
    x -= np.array([90.0, 98.0, 102.0], dtype="float32").reshape(1, 3, 1, 1, 1)
    return torch.from_numpy(x)


def synth_labels(B: int, seed: int, p_disrupt: float = 0.05) -> torch.Tensor:
    """int64 labels, class 0 = disruptive (src/dataset.py:91-94); both classes forced present."""
    import numpy as np
    rng = np.random.default_rng(seed + 7)
    y = (rng.random(B) >= p_disrupt).astype("int64")
    y[0] = 0
    if B > 1:
        y[1] = 1
    return torch.from_numpy(y)

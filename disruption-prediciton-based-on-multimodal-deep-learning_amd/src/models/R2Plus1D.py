"""R(2+1)D video classifier -- MI355X-native mirror of the reference's ``src/models/R2Plus1D.py``.

Same class names, constructor signatures, attribute names and state-dict keys as the reference
(``res2plus1d.conv1.spatio_conv.conv.weight`` ... ``linear.3.bias``), so checkpoints and the unchanged
training scripts work.  The nn.Conv3d / nn.BatchNorm3d children are PARAMETER HOLDERS only: the trunk
forward/backward is one call into the C++ executor (``md_plan_forward`` / ``md_plan_backward``) that
runs the hand-written gfx950 kernels; no ATen convolution / batch-norm kernel is ever launched.

Reference anchors: Conv3dBlock R2Plus1D.py:25-58, SpatioTemporalConv :115-162,
SpatioTemporalResBlock :164-187, SpatioTemporalResLayer :190-204, R2Plus1DNet :207-226,
R2Plus1DClassifier :228-288.
"""
from __future__ import annotations

import math
from typing import List, Tuple, Union

import torch
import torch.nn as nn
from torch.nn.modules.utils import _triple

from .. import _native as N
from .. import ops
from .._plan import TrunkFunction, TrunkPlan
from ._unit import conv_bn_leaky


class Conv3dBlock(nn.Module):
    """conv(bias) -> BatchNorm3d -> LeakyReLU(alpha)   (reference R2Plus1D.py:25-58)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size=3, stride=1, dilation: int = 1, padding=1,
                 bias: bool = False, alpha: float = 0.01):
        super().__init__()
        strides = stride if type(stride) == tuple else (1, stride, stride)
        kernel_sizes = kernel_size if type(kernel_size) == tuple else (1, kernel_size, kernel_size)
        paddings = padding if type(padding) == tuple else (0, padding, padding)
        if dilation != 1:
            raise NotImplementedError("mi355x hot path: dilation != 1 is not used by the reference configs")
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size=kernel_sizes, stride=strides, padding=paddings,
                              dilation=dilation, bias=bias)
        self.bn = nn.BatchNorm3d(out_channels)
        self.relu = nn.LeakyReLU(alpha)

    def forward(self, x: torch.Tensor):
        return conv_bn_leaky(x, self.conv, self.bn, self.relu.negative_slope, self.training)


class SpatioTemporalConv(nn.Module):
    """(1,k,k) block then (k,1,1) block (reference R2Plus1D.py:115-162)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size=(3, 1, 1), stride=(1, 1, 1), dilation: int = 1,
                 padding=(1, 1, 1), bias: bool = False, alpha: float = 0.01, is_first: bool = False):
        super().__init__()
        if type(kernel_size) == int:
            kernel_size = _triple(kernel_size)
        if type(stride) == int:
            stride = _triple(stride)
        if type(padding) == int:
            padding = _triple(padding)
        if is_first:
            mid = 45
            self.spatio_conv = Conv3dBlock(in_channels, mid, kernel_size, (1, stride[1], stride[2]), dilation, padding,
                                           False, alpha)
            self.temporal_conv = Conv3dBlock(mid, out_channels, (3, 1, 1), (stride[0], 1, 1), dilation, (1, 0, 0), False,
                                             alpha)
        else:
            kt, kh, kw = kernel_size
            mid = int(math.floor((kt * kh * kw * in_channels * out_channels) / (kh * kw * in_channels + kt * out_channels)))
            self.spatio_conv = Conv3dBlock(in_channels, mid, (1, kh, kw), (1, stride[1], stride[2]), dilation,
                                           (0, padding[1], padding[2]), bias, alpha)
            self.temporal_conv = Conv3dBlock(mid, out_channels, (kt, 1, 1), (stride[0], 1, 1), dilation,
                                             (padding[0], 0, 0), bias, alpha)

    def forward(self, x: torch.Tensor):
        return self.temporal_conv(self.spatio_conv(x))


class _AddLeakyFunction(torch.autograd.Function):
    """leaky_relu(x + res, alpha) as one native launch each way (md_add_leaky_fwd / _bwd): the residual close of a block used
    on its own (inside R2Plus1DNet the trunk executor closes blocks with md_residual_fwd on the raw tensors)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        ops.require_cuda(a, b)
        a = ops.f32(a).contiguous(); b = ops.f32(b).contiguous()
        out = torch.empty_like(a)
        N.check(N.lib().md_add_leaky_fwd(ops._p(a), ops._p(b), float(alpha), a.numel(), ops._p(out), ops._stream()), "md_add_leaky_fwd")
        ctx.save_for_backward(out)
        ctx.alpha = float(alpha)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        g = ops.f32(dout).contiguous()
        dx = torch.empty_like(out)
        N.check(N.lib().md_add_leaky_bwd(ops._p(out), ops._p(g), ctx.alpha, out.numel(), ops._p(dx), ops._stream()), "md_add_leaky_bwd")
        return dx, dx, None


class SpatioTemporalResBlock(nn.Module):
    """Reference R2Plus1D.py:164-187.  NB (quirk kept): the inner SpatioTemporalConv children are built
    without ``alpha`` and therefore use LeakyReLU(0.01); only the closing activation uses ``alpha``."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Union[Tuple[int, int, int], int] = (3, 1, 1),
                 downsample: bool = False, dilation: int = 1, alpha: float = 0.01):
        super().__init__()
        self.downsample = downsample
        padding = kernel_size // 2
        if self.downsample:
            self.downsample_conv = SpatioTemporalConv(in_channels, out_channels, kernel_size=1, stride=2,
                                                      dilation=dilation, padding=0)
            self.conv1 = SpatioTemporalConv(in_channels, out_channels, kernel_size, stride=(2, 2, 2), dilation=dilation,
                                            padding=padding)
        else:
            self.conv1 = SpatioTemporalConv(in_channels, out_channels, kernel_size, stride=(1, 1, 1), dilation=dilation,
                                            padding=padding)
        self.conv2 = SpatioTemporalConv(out_channels, out_channels, kernel_size, stride=(1, 1, 1), padding=padding,
                                        dilation=dilation)
        self.relu = nn.LeakyReLU(alpha)

    def forward(self, x: torch.Tensor):
        res = self.conv2(self.conv1(x))
        if self.downsample:
            x = self.downsample_conv(x)
        return _AddLeakyFunction.apply(x, res, self.relu.negative_slope)


class SpatioTemporalResLayer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: Union[Tuple[int, int, int], int] = (3, 1, 1),
                 downsample: bool = False, dilation: int = 1, alpha: float = 0.01, layer_size: int = 4):
        super().__init__()
        self.block1 = SpatioTemporalResBlock(in_channels, out_channels, kernel_size, downsample=downsample,
                                             dilation=dilation, alpha=alpha)
        self.blocks = nn.ModuleList([])
        for _ in range(layer_size - 1):
            self.blocks.append(SpatioTemporalResBlock(out_channels, out_channels, kernel_size, downsample=False,
                                                      dilation=dilation, alpha=alpha))

    def forward(self, x: torch.Tensor):
        x = self.block1(x)
        for block in self.blocks:
            x = block(x)
        return x


class R2Plus1DNet(nn.Module):
    """Stem + 4 residual stages + global average pool (reference R2Plus1D.py:207-226).

    forward() hands the whole trunk to the C++ executor: input (B,3,T,H,W) fp32 on the GPU -> (B,128).
    ``grad_segment_hook`` (optional) is called after the backward of each stage with the flat gradient
    buffer, which is how the data-parallel loop overlaps RCCL all-reduce with the rest of backward.
    """

    def __init__(self, layer_sizes: List[int] = [4, 4, 4, 4], alpha: float = 0.01):
        super().__init__()
        self.layer_sizes = [int(v) for v in layer_sizes]
        self.alpha = float(alpha)
        self.conv1 = SpatioTemporalConv(3, 32, kernel_size=(1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3), dilation=1,
                                        is_first=True, alpha=alpha)
        self.conv2 = SpatioTemporalResLayer(32, 32, 3, dilation=1, alpha=alpha, layer_size=layer_sizes[0])
        self.conv3 = SpatioTemporalResLayer(32, 64, 3, dilation=1, alpha=alpha, layer_size=layer_sizes[1], downsample=True)
        self.conv4 = SpatioTemporalResLayer(64, 64, 3, dilation=1, alpha=alpha, layer_size=layer_sizes[2], downsample=True)
        self.conv5 = SpatioTemporalResLayer(64, 128, 3, dilation=1, alpha=alpha, layer_size=layer_sizes[3], downsample=True)
        self.pool = nn.AdaptiveAvgPool3d(1)
        self._plans = {}
        self.grad_segment_hook = None

    def unit_modules(self) -> List[Conv3dBlock]:
        """Conv3dBlock instances in the executor's unit order."""
        units = [self.conv1.spatio_conv, self.conv1.temporal_conv]
        for layer in (self.conv2, self.conv3, self.conv4, self.conv5):
            for blk in [layer.block1] + list(layer.blocks):
                units += [blk.conv1.spatio_conv, blk.conv1.temporal_conv, blk.conv2.spatio_conv, blk.conv2.temporal_conv]
                if blk.downsample:
                    units += [blk.downsample_conv.spatio_conv, blk.downsample_conv.temporal_conv]
        return units

    def _plan(self, B, T, H, W) -> TrunkPlan:
        key = (B, T, H, W)
        p = self._plans.get(key)
        if p is None:
            p = TrunkPlan(B, T, H, W, self.layer_sizes, self.alpha)
            units = self.unit_modules()
            if len(units) != p.num_units:
                raise RuntimeError("module tree and executor plan disagree on the number of conv units")
            for i, u in enumerate(units):
                if tuple(u.conv.weight.shape) != p.weight_shape(i):
                    raise RuntimeError(f"unit {i}: weight {tuple(u.conv.weight.shape)} != plan {p.weight_shape(i)}")
            self._plans[key] = p
        return p

    @staticmethod
    def output_channels() -> int:
        return 128

    def forward(self, x: torch.Tensor):
        if x.dim() != 5 or x.size(1) != 3:
            raise RuntimeError(f"R2Plus1DNet expects (B,3,T,H,W), got {tuple(x.shape)}")
        ops.require_cuda(x.contiguous())
        x = x.contiguous().float()
        B, _, T, H, W = x.shape
        plan = self._plan(B, T, H, W)
        units = self.unit_modules()
        ws = [u.conv.weight for u in units]
        gs = [u.bn.weight for u in units]
        bs = [u.bn.bias for u in units]
        rms = [u.bn.running_mean for u in units]
        rvs = [u.bn.running_var for u in units]
        # a backward can follow only if grad mode is on and something upstream of the features requires a gradient
        ops.check_fp16_range(x, "the clip")
        need_bwd = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in ws + gs + bs))
        feat = TrunkFunction.apply(plan, x, rms, rvs, self.training, need_bwd, self.grad_segment_hook, *ws, *gs, *bs)
        if self.training:
            torch._foreach_add_([u.bn.num_batches_tracked for u in units], 1)
        return feat


class _Head(nn.Sequential):
    """Linear -> BatchNorm1d -> ELU -> Linear (reference R2Plus1D.py:243-248) as ONE fused HIP kernel pair."""

    def forward(self, f: torch.Tensor):
        from ._unit import head_apply
        lin0, bn, elu, lin1 = self[0], self[1], self[2], self[3]
        return head_apply(f, lin0, bn, lin1, float(elu.alpha), self.training)


class R2Plus1DClassifier(nn.Module):
    def __init__(self, input_size: Tuple[int, int, int, int] = (3, 8, 112, 112), num_classes: int = 2,
                 layer_sizes: List[int] = [4, 4, 4, 4], pretrained: bool = False, alpha: float = 1.0):
        super().__init__()
        self.input_size = input_size
        self.res2plus1d = R2Plus1DNet(layer_sizes, alpha=alpha)
        linear_dims = self.get_res2plus1d_output_size()[1]
        self.linear = _Head(
            nn.Linear(linear_dims, linear_dims // 2),
            nn.BatchNorm1d(linear_dims // 2),
            nn.ELU(alpha),
            nn.Linear(linear_dims // 2, num_classes),
        )
        self.__init_weight()
        if pretrained:
            self.__load_pretrained_weights()

    def get_res2plus1d_output_size(self):
        # The reference sizes the head with a dummy CPU forward (R2Plus1D.py:255-259); the pooled trunk
        # output is (1, 128) for every input size, so no forward is needed (and none is possible on CPU).
        return torch.Size((1, R2Plus1DNet.output_channels()))

    def __load_pretrained_weights(self):
        s_dict = self.state_dict()
        for name in s_dict:
            print(name)
            print(s_dict[name].size())

    def __init_weight(self):
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm3d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def encode(self, x: torch.Tensor):
        with torch.no_grad():
            x = self.res2plus1d(x)
        return x

    def forward(self, x: torch.Tensor):
        x = self.res2plus1d(x)
        x = self.linear(x)
        return x

    def summary(self, device: str = 'cpu', show_input: bool = True, show_hierarchical: bool = True,
                print_summary: bool = False, show_parent_layers: bool = False):
        n = sum(p.numel() for p in self.parameters())
        lines = [f"R2Plus1DClassifier(input={tuple(self.input_size)}, layer_sizes={self.res2plus1d.layer_sizes}) "
                 f"-- {n:,} parameters (MI355X executor; per-layer table omitted)"]
        for i, u in enumerate(self.res2plus1d.unit_modules()):
            lines.append(f"  unit {i:2d}: conv{tuple(u.conv.weight.shape)} stride={u.conv.stride} pad={u.conv.padding}")
        return print("\n".join(lines))

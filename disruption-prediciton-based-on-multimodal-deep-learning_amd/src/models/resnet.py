"""MI355X-native pieces of the reference's ``src/models/resnet.py``.

Built here: ``SwishEfficient`` / ``Swish`` (resnet.py:63-81) on ``md_swish_fwd`` / ``md_swish_bwd`` , ``Bottleneck3D``
(resnet.py:121-200) and the ``ResNet3D`` base (resnet.py:202-273).  The rest of that module
(BasicBlock3D, SubBatchNorm3d, Bottleneck2DPlus1D, ...) is dead code in the reference (never
constructed by any configuration) and is not rebuilt; this module never loads or executes reference source.
"""

import torch
import torch.nn as nn

from .. import _native as N
from .. import ops
from ._unit import CLAct, _AbsorbedBias, bump_batches_tracked, conv_bn_leaky, from_cl_act, to_cl_act


class SwishEfficient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ops.require_cuda(x)
        x = ops.f32(x).contiguous()
        y = torch.empty_like(x)
        N.check(N.lib().md_swish_fwd(ops._p(x), x.numel(), ops._p(y), ops._stream()), "md_swish_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        g = ops.f32(grad_output).contiguous()
        dx = torch.empty_like(x)
        N.check(N.lib().md_swish_bwd(ops._p(x), ops._p(g), x.numel(), ops._p(dx), ops._stream()), "md_swish_bwd")
        return dx


class Swish(nn.Module):
    def __init__(self):
        super(Swish, self).__init__()

    def forward(self, x):
        return SwishEfficient.apply(x)


class _SESwishFunction(torch.autograd.Function):
    """out = swish(a * sigmoid(fc2(relu(fc1(mean_thw a)))))   (Bottleneck3D.forward, resnet.py:182-190)."""

    @staticmethod
    def forward(ctx, a, w1, b1, w2, b2):
        ops.require_cuda(a, w1, b1, w2, b2)
        a = ops.f32(a).contiguous()
        Nn, Cc = a.shape[0], a.shape[1]
        thw = a.numel() // (Nn * Cc)
        Wd = w1.shape[0]
        w1m = w1.reshape(Wd, Cc).contiguous(); w2m = w2.reshape(Cc, Wd).contiguous()
        pool = torch.empty(Nn, Cc, device=a.device); hidden = torch.empty(Nn, Wd, device=a.device)
        gate = torch.empty(Nn, Cc, device=a.device); out = torch.empty_like(a)
        N.check(N.lib().md_se_swish_fwd(ops._p(a), Nn, Cc, thw, Wd, ops._p(w1m), ops._p(b1.contiguous()), ops._p(w2m),
                                        ops._p(b2.contiguous()), ops._p(pool), ops._p(hidden), ops._p(gate), ops._p(out),
                                        ops._stream()), "md_se_swish_fwd")
        ctx.save_for_backward(a, w1m, w2m, pool, hidden, gate)
        ctx.wshapes = (w1.shape, w2.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, w1m, w2m, pool, hidden, gate = ctx.saved_tensors
        Nn, Cc = a.shape[0], a.shape[1]
        thw = a.numel() // (Nn * Cc)
        Wd = w1m.shape[0]
        g = ops.f32(dout).contiguous()
        da = torch.empty_like(a)
        dw1 = torch.empty_like(w1m); db1 = torch.empty(Wd, device=a.device)
        dw2 = torch.empty_like(w2m); db2 = torch.empty(Cc, device=a.device)
        scratch = torch.empty(3 * Nn * Cc + Nn * Wd, device=a.device)
        N.check(N.lib().md_se_swish_bwd(ops._p(a), ops._p(g), Nn, Cc, thw, Wd, ops._p(w1m), ops._p(w2m), ops._p(pool),
                                        ops._p(hidden), ops._p(gate), ops._p(da), ops._p(dw1), ops._p(db1), ops._p(dw2),
                                        ops._p(db2), ops._p(scratch), ops._stream()), "md_se_swish_bwd")
        s1, s2 = ctx.wshapes
        return da, dw1.reshape(s1), db1, dw2.reshape(s2), db2


class _AddReluFunction(torch.autograd.Function):
    """relu(a + b): the residual close (resnet.py:196-198)."""

    @staticmethod
    def forward(ctx, a, b):
        ops.require_cuda(a, b)
        a = ops.f32(a).contiguous(); b = ops.f32(b).contiguous()
        out = torch.empty_like(a)
        N.check(N.lib().md_add_relu_fwd(ops._p(a), ops._p(b), a.numel(), ops._p(out), ops._stream()), "md_add_relu_fwd")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        g = ops.f32(dout).contiguous()
        dx = torch.empty_like(out)
        N.check(N.lib().md_add_relu_bwd(ops._p(out), ops._p(g), out.numel(), ops._p(dx), ops._stream()), "md_add_relu_bwd")
        return dx, dx


class Bottleneck3D(nn.Module):
    """MI355X-native mirror of the reference's Bottleneck3D (resnet.py:121-200): same constructor, child-module names and
    state-dict keys; every convolution + BatchNorm (+ReLU) runs as one gfx950 unit (``_unit.conv_bn_leaky`` with slope 0 /
    1), the squeeze-excitation gate with Swish and the residual close as fused kernels.  ``base_bn_splits`` (SubBatchNorm3d)
    and ``bias=True`` are not built."""
    expansion = 4

    def __init__(self, in_planes: int, planes: int, stride: int = 1, downsample=None, bias: bool = False, head_conv: int = 1,
                 base_bn_splits=None, index: int = 0):
        super(Bottleneck3D, self).__init__()
        if base_bn_splits is not None:
            raise NotImplementedError("mi355x hot path: SubBatchNorm3d (base_bn_splits) is not built")
        if bias:
            raise NotImplementedError("mi355x hot path: Bottleneck3D with bias=True is not used by the reference")
        self.index = index
        if head_conv == 1:
            self.conv1 = nn.Conv3d(in_planes, planes, kernel_size=1, bias=False)
        elif head_conv == 3:
            self.conv1 = nn.Conv3d(in_planes, planes, kernel_size=(3, 1, 1), bias=False, padding=(1, 0, 0))
        else:
            raise ValueError("Unsupported head_conv!")
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = nn.Conv3d(planes, planes, kernel_size=(1, 3, 3), stride=(1, stride, stride), padding=(0, 1, 1), bias=bias)
        self.bn2 = nn.BatchNorm3d(planes)
        self.conv3 = nn.Conv3d(planes, planes * 4, kernel_size=1, bias=bias)
        self.bn3 = nn.BatchNorm3d(planes * 4)
        self.swish = Swish()
        self.relu = nn.ReLU(inplace=True)
        if self.index % 2 == 0:
            width = self.round_width(planes)
            self.global_pool = nn.AdaptiveAvgPool3d((1, 1, 1))
            self.fc1 = nn.Conv3d(planes, width, kernel_size=1, stride=1)
            self.fc2 = nn.Conv3d(width, planes, kernel_size=1, stride=1)
            self.sigmoid = nn.Sigmoid()
        self.downsample = downsample
        self.stride = stride

    def round_width(self, width: int, multiplier=0.0625, min_width=8, divisor=8):
        if not multiplier:
            return width
        width *= multiplier
        min_width = min_width or divisor
        width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
        if width_out < 0.9 * width:
            width_out += divisor
        return int(width_out)

    def _downsample(self, x):
        ds = self.downsample
        if (isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv3d) and isinstance(ds[1], nn.BatchNorm3d)
                and ds[0].bias is None):
            return conv_bn_leaky(x, ds[0], ds[1], 1.0, self.training)          # conv + BN, no activation (resnet.py:254-257)
        raise NotImplementedError("mi355x hot path: downsample must be Sequential(Conv3d(bias=False), BatchNorm3d)")

    def forward(self, x):
        """x: a (B,C,T,H,W) tensor, or a CLAct (the channels-last activation the native stages pass between blocks: then every
        unit, the Swish and the residual close work on that layout and a CLAct comes back; only the squeeze-excitation gate,
        whose pooling kernel wants channel planes, converts there and back)."""
        cl = isinstance(x, CLAct)
        out = conv_bn_leaky(x, self.conv1, self.bn1, 0.0, self.training)       # conv1 -> bn1 -> relu
        out = conv_bn_leaky(out, self.conv2, self.bn2, 0.0, self.training)     # conv2 -> bn2 -> relu
        if self.index % 2 == 0:
            a = from_cl_act(out) if cl else out
            a = _SESwishFunction.apply(a, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
            out = to_cl_act(a) if cl else a
        elif cl:
            out = CLAct(SwishEfficient.apply(out.t), out.C)                    # elementwise, swish(0) = 0 keeps the padding
        else:
            out = SwishEfficient.apply(out)
        out = conv_bn_leaky(out, self.conv3, self.bn3, 1.0, self.training)     # conv3 -> bn3
        residual = x if self.downsample is None else self._downsample(x)
        if cl:
            return CLAct(_AddReluFunction.apply(out.t, residual.t), out.C)
        return _AddReluFunction.apply(out, residual)


def _conv_bias_bn_relu(x, conv: nn.Conv3d, bn: nn.BatchNorm3d, training: bool):
    """Conv3d(bias=True) -> BatchNorm3d -> ReLU (ResNet3D.layer0, resnet.py:221-224) on the bias-free gfx950 unit: the bias
    only shifts the batch mean, so it is folded into the running mean (kept identical to the reference's, state dicts
    stay interchangeable) and gets its exact zero gradient."""
    if conv.bias is None:
        return conv_bn_leaky(x, conv, bn, 0.0, training)
    from ._unit import ConvBnLeakyFunction
    b = conv.bias.detach()
    rmean = bn.running_mean if training else bn.running_mean - b              # eval: (y + b - rm) == (y - (rm - b))
    out = ConvBnLeakyFunction.apply(x, conv.weight, bn.weight, bn.bias, rmean, bn.running_var, conv.stride, conv.padding, 0.0,
                                    bool(training), float(bn.eps), float(bn.momentum))
    if training:
        bn.running_mean.add_(b * bn.momentum)                                  # the batch mean of (y + b) is mean(y) + b
        bump_batches_tracked(bn)
        out = _AbsorbedBias.apply(out, conv.bias)
    return out


class ResNet3D(nn.Module):
    """Mirror of the reference's ResNet3D base (resnet.py:202-273): same constructor keywords (in_channels, alpha, slow,
    base_bn_splits), same children (layer0 .. layer4).  ``forward`` is left to SlowNet / FastNet as in the reference; the
    stem runs through ``stem()`` (conv+BN+ReLU unit, then the (1,3,3)/2 max pool kernel)."""

    def __init__(self, block, layers, **kwargs):
        super(ResNet3D, self).__init__()
        in_channels = kwargs['in_channels']
        self.alpha = kwargs['alpha']
        self.slow = kwargs['slow']  # slow->1 else fast->0
        m = 16
        self.inplanes = (m + m // self.alpha) if self.slow else m // self.alpha
        self.base_bn_splits = kwargs["base_bn_splits"]
        out_channels = m // (1 if self.slow else self.alpha)
        self.layer0 = nn.Sequential(
            nn.Conv3d(in_channels, out_channels, kernel_size=(1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3)),
            nn.BatchNorm3d(out_channels),
            nn.ReLU(inplace=True),
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        )
        self.layer1 = self._make_layer(block, m // (1 if self.slow else self.alpha), layers[0],
                                       head_conv=1 if self.slow else 3, base_bn_splits=self.base_bn_splits)
        self.layer2 = self._make_layer(block, 2 * m // (1 if self.slow else self.alpha), layers[1], stride=2,
                                       head_conv=1 if self.slow else 3, base_bn_splits=self.base_bn_splits)
        self.layer3 = self._make_layer(block, 4 * m // (1 if self.slow else self.alpha), layers[2], stride=2,
                                       head_conv=3, base_bn_splits=self.base_bn_splits)
        self.layer4 = self._make_layer(block, 8 * m // (1 if self.slow else self.alpha), layers[3], stride=2,
                                       head_conv=3, base_bn_splits=self.base_bn_splits)

    def init_params(self):
        import torch.nn.init as nn_init
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn_init.xavier_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm3d) and m.weight is not None:
                nn_init.constant_(m.weight, 1)

    def stem(self, x):
        from ._unit import MaxPool1x3x3Function
        x = _conv_bias_bn_relu(x, self.layer0[0], self.layer0[1], self.training)
        return MaxPool1x3x3Function.apply(x)

    def forward(self, x):
        raise NotImplementedError('use each pathway network\' forward function')

    def _make_layer(self, block, planes: int, blocks: int = 3, stride: int = 1, head_conv: int = 1, base_bn_splits=None):
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv3d(self.inplanes, planes * block.expansion, kernel_size=1, stride=(1, stride, stride), bias=False),
                nn.BatchNorm3d(planes * block.expansion)
            )
        else:
            downsample = None
        layers = list()
        layers.append(block(self.inplanes, planes, stride, downsample, head_conv=head_conv, base_bn_splits=base_bn_splits))
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            layers.append(block(self.inplanes, planes, head_conv=head_conv, base_bn_splits=base_bn_splits))
        self.inplanes += self.slow * block.expansion * planes // self.alpha
        return nn.Sequential(*layers)

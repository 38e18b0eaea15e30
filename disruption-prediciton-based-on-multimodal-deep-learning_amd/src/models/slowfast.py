"""MI355X-native mirror of the reference's ``src/models/slowfast.py`` (same classes, constructor arguments, child-module
names and state-dict keys: ``encoder.slownet.layer4[0].downsample[0]``, ``encoder.fastnet.l_layer3``, ``classifier.classifier``).

Every convolution + BatchNorm runs as a gfx950 unit (Bottleneck3D / ResNet3D in ``resnet.py``), the laterals as plain
convolutions, the pools and the classifier head as fused kernels.  ``torch.cat`` and the temporal sub-sampling
``x[:, :, ::tau]`` are memory plumbing and stay torch calls.  Differences a maintainer should know: inputs must be CUDA
tensors (no CPU dummy forward: the classifier width is computed from the architecture, 128*m/alpha... = 512 + 128 = 640 for
the defaults, instead of ``get_output_shape()`` tracing a CPU sample, slowfast.py:186-187).
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .resnet import *          # noqa: F401,F403  (as the reference does, slowfast.py:5)
from .resnet import Bottleneck3D, ResNet3D
import os

from ..utils import streams
from ._unit import GlobalAvgPoolFunction, cat_cl, conv_plain, deferred_bn_counters, from_cl_act, head_apply, to_cl_act

# the stages keep their activations in the kernels' channels-last layout (MD_SLOWFAST_CL=0: reference layout at every unit boundary)
_CL = not (os.environ.get("MD_SLOWFAST_CL") == "0")


_STAGES = ("layer1", "layer2", "layer3", "layer4")
_LATERALS = ("l_maxpool", "l_layer1", "l_layer2", "l_layer3")          # one per point where the fast path feeds the slow one


class SlowNet(ResNet3D):
    """Slow pathway: before every stage the lateral tensor of the fast pathway is concatenated on the channel axis."""

    def __init__(self, blocks, layers, **kwargs):
        super(SlowNet, self).__init__(blocks, layers, **kwargs)
        self.init_params()

    def forward(self, x: Tuple[torch.Tensor, List[torch.Tensor]], ready=None):
        """``ready``: one event per lateral when the fast pathway runs on a side stream (this pathway waits for lateral k only
        when stage k needs it)."""
        h, laterals = x
        h = self.stem(h)
        if _CL:
            h = to_cl_act(h)
        for k, (name, lat) in enumerate(zip(_STAGES, laterals)):
            if ready is not None:
                streams.wait(ready[k], lat)
            h = getattr(self, name)(cat_cl(h, lat) if _CL else torch.cat([h, lat], dim=1))
        return GlobalAvgPoolFunction.apply(from_cl_act(h) if _CL else h)


def resnet50_s(block=Bottleneck3D, layers=[3, 4, 6, 3], **kwargs):
    return SlowNet(block, layers, **kwargs)


class FastNet(ResNet3D):
    """Fast pathway plus the four time-strided lateral convolutions (kernel (alpha+2,1,1), stride (alpha,1,1), no bias) that
    bring its stem / stage outputs to the slow pathway's frame rate."""

    def __init__(self, blocks, layers, **kwargs):
        super(FastNet, self).__init__(blocks, layers, **kwargs)
        a = kwargs["alpha"]
        for name, mult in zip(_LATERALS, (1, 4, 8, 16)):
            width = mult * 16 // self.alpha
            setattr(self, name, nn.Conv3d(width, width, kernel_size=(a + 2, 1, 1), stride=(a, 1, 1), padding=(1, 0, 0), bias=False))
        self.init_params()

    def forward(self, x: torch.Tensor, mark=None):
        """``mark()`` (optional) is called after every lateral and returns an event; the events come back as the third result."""
        h = self.stem(x)
        if _CL:
            h = to_cl_act(h)
        laterals = [conv_plain(h, self.l_maxpool)]
        ready = [mark()] if mark is not None else None
        for name, lat in zip(_STAGES[:3], _LATERALS[1:]):
            h = getattr(self, name)(h)
            laterals.append(conv_plain(h, getattr(self, lat)))
            if mark is not None:
                ready.append(mark())
        h = self.layer4(h)
        if mark is not None:
            return GlobalAvgPoolFunction.apply(from_cl_act(h) if _CL else h), laterals, ready
        return GlobalAvgPoolFunction.apply(from_cl_act(h) if _CL else h), laterals


def resnet50_f(block=Bottleneck3D, layers=[3, 4, 6, 3], **kwargs):
    return FastNet(block, layers, **kwargs)


class SlowFastEncoder(nn.Module):
    def __init__(self, input_shape: Tuple[int, int, int, int] = (3, 8, 112, 112), block=Bottleneck3D,
                 layers: List[int] = [3, 4, 6, 3], alpha: int = 4, tau_fast: int = 1):
        super(SlowFastEncoder, self).__init__()
        self.input_shape = input_shape
        self.seq_len = input_shape[1]
        self.in_channels = input_shape[0]
        self.alpha = alpha
        self.tau_fast = tau_fast
        self.slownet = resnet50_s(block=block, layers=layers, alpha=alpha, in_channels=self.in_channels, slow=1,
                                  base_bn_splits=None)
        self.fastnet = resnet50_f(block=block, layers=layers, alpha=alpha, in_channels=self.in_channels, slow=0,
                                  base_bn_splits=None)
        self._out_dim = 8 * 16 * block.expansion + 8 * 16 // alpha * block.expansion       # slow + fast pooled widths

    def split_slow_fast(self, x: torch.Tensor):
        tau_fast = self.tau_fast
        tau_slow = tau_fast * self.alpha
        return x[:, :, ::tau_slow, :, :], x[:, :, ::tau_fast, :, :]

    def forward(self, x: torch.Tensor):
        from .. import ops
        with deferred_bn_counters(), ops.prepacked(self):   # one launch for all BatchNorm step counters / (almost) all weight packs
            x_slow, x_fast = self.split_slow_fast(x)
            if streams.enabled(x):
                # the fast pathway on a side stream; the slow one follows on the current stream and waits lateral by lateral
                with streams.fork(x.device, 0, (x,)) as f:
                    x_fast, laterals, ready = self.fastnet(x_fast, f.mark)
                x_slow = self.slownet((x_slow, laterals), ready)
                f.join(x_fast)
            else:
                x_fast, laterals = self.fastnet(x_fast)
                x_slow = self.slownet((x_slow, laterals))
            return torch.cat([x_slow, x_fast], dim=1)

    def show_CAM(self):
        pass

    def show_Grad_CAM(self):
        pass

    def get_output_shape(self):
        return torch.Size((1, self._out_dim))


class SlowFastClassifier(nn.Module):
    def __init__(self, input_dim: int, num_classes: int = 2, alpha: float = 1.0):
        super(SlowFastClassifier, self).__init__()
        self.input_dim = input_dim
        self.classifier = nn.Sequential(
            nn.Linear(input_dim, input_dim // 2),
            nn.BatchNorm1d(input_dim // 2),
            nn.ELU(alpha),
            nn.Linear(input_dim // 2, num_classes)
        )

    def forward(self, x: torch.Tensor):
        lin0, bn, elu, lin1 = self.classifier[0], self.classifier[1], self.classifier[2], self.classifier[3]
        return head_apply(x, lin0, bn, lin1, float(elu.alpha), bool(self.training))


class SlowFast(nn.Module):
    def __init__(self, input_shape: Tuple[int, int, int, int] = (3, 8, 112, 112), block=Bottleneck3D,
                 layers: List[int] = [3, 4, 6, 3], alpha: int = 4, tau_fast: int = 1, num_classes: int = 2,
                 alpha_elu: float = 1.0):
        super(SlowFast, self).__init__()
        self.input_shape = input_shape
        self.encoder = SlowFastEncoder(input_shape, block, layers, alpha, tau_fast)
        cls_input_dim = self.encoder.get_output_shape()[-1]
        self.classifier = SlowFastClassifier(cls_input_dim, num_classes, alpha_elu)

    def encode(self, x: torch.Tensor):
        with torch.no_grad():
            x = self.encoder.forward(x)
            return x.view(x.size()[0], -1)

    def forward(self, x: torch.Tensor):
        x = self.encoder.forward(x)
        return self.classifier.forward(x)

    def summary(self, device: str = 'cpu', show_input: bool = True, show_hierarchical: bool = True, print_summary: bool = False,
                show_parent_layers: bool = True):
        rows = ["%-60s %-24s %d" % (k, tuple(v.shape), v.numel()) for k, v in self.named_parameters()]
        text = "\\n".join(rows + ["total parameters: %d" % sum(p.numel() for p in self.parameters())])
        if print_summary:
            print(text)
        return text

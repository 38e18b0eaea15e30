"""MI355X-native mirror of the reference's ``src/models/NoiseLayer.py`` (same class, same constructor).

Training: ``x + (mean + randn * std)``.  The reference draws the noise with the CPU default generator
(``torch.randn(x.size())``, NoiseLayer.py:13) and moves it to the input's device; this mirror keeps exactly that, so
seeded runs see the same noise, and does the arithmetic in one launch (``md_add_noise``).  Eval: identity.
The 0D encoders of the reference (CnnLSTM.py:38, MLSTM_FCN.py:113, transformer.py:57) import this module by name.
"""
import torch
import torch.nn as nn

from .. import _native as N
from .. import ops


class _AddNoise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, noise, mean, std):
        out = torch.empty_like(x)
        N.check(N.lib().md_add_noise(ops._p(x), ops._p(noise), float(mean), float(std), x.numel(), ops._p(out), ops._stream()),
                "md_add_noise")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return grad_out, None, None, None


class NoiseLayer(nn.Module):
    def __init__(self, mean: float = 0, std: float = 1e-2):
        super().__init__()
        self.mean = mean
        self.std = std

    def forward(self, x: torch.Tensor):
        if not self.training:
            return x
        ops.require_cuda(x)
        noise = torch.randn(x.size()).to(x.device)              # CPU generator, as the reference
        return _AddNoise.apply(ops.f32(x), noise, self.mean, self.std)

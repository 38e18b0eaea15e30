"""CPU restatement (test infrastructure only) of the reference's elementwise modules on the next rows of the scope
table.  Pinned by tests/golden/elementwise.npz (generated from the reference, tests/golden/make_golden.py).

  * swish / swish_backward  -- SwishEfficient, src/models/resnet.py:70-81
  * noise_layer             -- NoiseLayer.forward, src/models/NoiseLayer.py:11-16 (training: noise drawn with the CPU
                               default generator, ``torch.randn(x.size())``; eval: identity)
"""
import torch


def swish(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(x)                                   # resnet.py:73


def swish_backward(x: torch.Tensor, grad_output: torch.Tensor) -> torch.Tensor:
    s = torch.sigmoid(x)                                          # resnet.py:80
    return grad_output * (s * (1 + x * (1 - s)))                  # resnet.py:81


def noise_layer(x: torch.Tensor, mean: float, std: float, training: bool) -> torch.Tensor:
    if not training:
        return x                                                  # NoiseLayer.py:16
    noise = torch.ones_like(x) * mean + torch.randn(x.size()) * std   # NoiseLayer.py:13 (CPU generator)
    return x + noise

"""Rows a8 / a15 of the scope table: SwishEfficient and NoiseLayer.
CPU: the oracle restatement against the fixture recorded from the reference (bit-exact: same torch ops).
GPU: the HIP kernels through the mirrored modules against the same fixture -- Swish within 2e-6 of the output scale
(expf vs ATen's vectorised exp), its gradient likewise; NoiseLayer bit-exact in eval mode and within 1 ulp-level
(1e-7 relative) in training mode with the reference's seed (same CPU draws, one fused add instead of two)."""
import os

import numpy as np
import pytest
import torch

from oracle import elementwise as oe


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "elementwise.npz"))


def test_oracle_matches_reference_fixture(fx):
    x = torch.from_numpy(fx["swish_x"]); dy = torch.from_numpy(fx["swish_dy"])
    assert np.array_equal(oe.swish(x).numpy(), fx["swish_y"])
    assert np.array_equal(oe.swish_backward(x, dy).numpy(), fx["swish_dx"])
    xn = torch.from_numpy(fx["noise_x"])
    torch.manual_seed(int(fx["noise_seed"]))
    assert np.array_equal(oe.noise_layer(xn, float(fx["noise_mean"]), float(fx["noise_std"]), True).numpy(), fx["noise_train"])
    assert np.array_equal(oe.noise_layer(xn, 0.0, 1.0, False).numpy(), fx["noise_eval"])


@pytest.mark.gpu
def test_swish_kernels_match_reference_fixture(fx):
    from src.models.resnet import Swish, SwishEfficient
    x = torch.from_numpy(fx["swish_x"]).cuda().requires_grad_(True)
    y = Swish()(x)
    y.backward(torch.from_numpy(fx["swish_dy"]).cuda())
    sy = max(1.0, float(np.abs(fx["swish_y"]).max())); sd = max(1.0, float(np.abs(fx["swish_dx"]).max()))
    assert float(np.abs(y.detach().cpu().numpy() - fx["swish_y"]).max()) <= 2e-6 * sy
    assert float(np.abs(x.grad.cpu().numpy() - fx["swish_dx"]).max()) <= 2e-6 * sd
    # ragged length / unaligned start, and a 5-D activation as Bottleneck3D passes it
    z = torch.randn(3, 5, 2, 7, 9, device="cuda")
    assert torch.allclose(SwishEfficient.apply(z[:, 1:].contiguous()), oe.swish(z[:, 1:].cpu()).cuda(), atol=2e-6, rtol=2e-6)
    with pytest.raises(RuntimeError):
        SwishEfficient.apply(torch.randn(4))


@pytest.mark.gpu
def test_noise_layer_matches_reference_fixture(fx):
    from src.models.NoiseLayer import NoiseLayer
    layer = NoiseLayer(mean=float(fx["noise_mean"]), std=float(fx["noise_std"]))
    xn = torch.from_numpy(fx["noise_x"]).cuda().requires_grad_(True)
    layer.train()
    torch.manual_seed(int(fx["noise_seed"]))
    out = layer(xn)
    assert float(np.abs(out.detach().cpu().numpy() - fx["noise_train"]).max()) <= 1e-7 * float(np.abs(fx["noise_train"]).max())
    out.sum().backward()
    assert torch.equal(xn.grad, torch.ones_like(xn))
    layer.eval()
    assert np.array_equal(layer(xn).detach().cpu().numpy(), fx["noise_eval"])


@pytest.mark.gpu
def test_fp16_range_check_is_loud_when_enabled(monkeypatch):
    """The split-fp16 forward needs |x| < 65504 (INTEGRATION 'Value range').  With MD_CHECK_RANGE=1 an input
    beyond that raises instead of returning inf; without it the same call returns non-finite values (documented limit)."""
    from src import ops
    from src.models._unit import linear_wb
    x = torch.full((8, 16), 1.0e5, device="cuda"); w = torch.randn(4, 16, device="cuda"); b = torch.zeros(4, device="cuda")
    monkeypatch.setattr(ops, "CHECK_RANGE", True)
    with pytest.raises(RuntimeError, match="65504"):
        linear_wb(x, w, b)
    monkeypatch.setattr(ops, "CHECK_RANGE", False)
    assert not bool(torch.isfinite(linear_wb(x, w, b)).all())
    assert bool(torch.isfinite(linear_wb(x * 1e-2, w, b)).all())

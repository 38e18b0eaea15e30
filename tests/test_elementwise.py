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


def test_resnet_module_adopts_the_reference_for_unbuilt_names(tmp_path):
    """With MD_REFERENCE_SRC set, src.models.resnet re-exports the reference's other classes with their Swish replaced by
    the native one (host logic only; needs the reference checkout, so it runs in the build container only)."""
    import subprocess, sys
    ref = os.environ.get("REFERENCE_ROOT", "/root/reference")
    if not os.path.isfile(os.path.join(ref, "src", "models", "resnet.py")):
        pytest.skip("reference checkout not present")
    (tmp_path / "pytorch_model_summary.py").write_text("def summary(*a, **k):\n    return ''\n")      # absent here, unused
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "disruption-prediciton-based-on-multimodal-deep-learning_amd")
    code = ("import src.models.resnet as r\n"
            "b = r.Bottleneck3D(16, 4, index=0)\n"
            "assert type(b.swish) is r.Swish and r.Swish.__module__ == 'src.models.resnet'\n"
            "assert r.Bottleneck3D.__module__ == 'src.models.resnet' and hasattr(r, 'ResNet3D')\n"
            "assert r.Swish().forward.__func__.__globals__['SwishEfficient'] is r.SwishEfficient\n"
            "print('ok', len([n for n in dir(r) if not n.startswith('_')]))\n")
    env = dict(os.environ, MD_REFERENCE_SRC=os.path.join(ref, "src"), PYTHONPATH=os.pathsep.join([pkg, str(tmp_path)]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stderr[-2000:]

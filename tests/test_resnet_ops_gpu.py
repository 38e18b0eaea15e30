"""GPU: the pieces of ResNet3D / SlowFast outside the bottleneck blocks against plain PyTorch on the CPU:
MaxPool3d((1,3,3),(1,2,2),(0,1,1)) incl. ties / odd sizes (forward values and argmax-routing of the gradient bit-exact),
global average pooling (1e-6), and the plain lateral convolutions (k=(alpha+2,1,1), stride (alpha,1,1), pad (1,0,0); 3e-5
forward, 5e-5 gradients, both arithmetic modes)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src import ops
    from src.models._unit import ConvFunction, GlobalAvgPoolFunction, MaxPool1x3x3Function


@pytest.mark.parametrize("shape", [(2, 3, 2, 8, 8), (1, 5, 3, 7, 9), (2, 2, 1, 1, 1), (1, 4, 2, 16, 5)])
def test_maxpool_matches_torch(shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    x[0, 0, 0] = torch.round(x[0, 0, 0])            # ties: the first maximum must take the gradient
    xr = x.clone().requires_grad_(True)
    y = F.max_pool3d(xr, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xg = x.cuda().requires_grad_(True)
    yg = MaxPool1x3x3Function.apply(xg)
    yg.backward(dy.cuda())
    assert torch.equal(yg.detach().cpu(), y.detach())
    assert torch.equal(xg.grad.cpu(), xr.grad)


def test_global_avg_pool_matches_torch():
    x = torch.randn(3, 7, 4, 5, 6)
    xr = x.clone().requires_grad_(True)
    y = F.adaptive_avg_pool3d(xr, 1).view(-1, 7)
    dy = torch.randn(3, 7)
    y.backward(dy)
    xg = x.cuda().requires_grad_(True)
    yg = GlobalAvgPoolFunction.apply(xg)
    yg.backward(dy.cuda())
    assert float((yg.detach().cpu() - y.detach()).abs().max()) < 1e-6
    assert float((xg.grad.cpu() - xr.grad).abs().max()) < 1e-7


@pytest.mark.parametrize("exact", [False, True], ids=["split", "exact_fp32"])
@pytest.mark.parametrize("alpha,C,T", [(4, 8, 16), (2, 16, 9), (8, 4, 32)])
def test_lateral_conv_matches_torch(alpha, C, T, exact):
    g = torch.Generator().manual_seed(alpha * 100 + C)
    x = torch.randn(2, C, T, 6, 5, generator=g)
    w = torch.randn(C, C, alpha + 2, 1, 1, generator=g) / (C * (alpha + 2)) ** 0.5
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    y = F.conv3d(xr, wr, None, (alpha, 1, 1), (1, 0, 0))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ops.set_exact_fp32(exact)
    try:
        xg = x.cuda().requires_grad_(True); wg = w.cuda().requires_grad_(True)
        yg = ConvFunction.apply(xg, wg, (alpha, 1, 1), (1, 0, 0))
        yg.backward(dy.cuda())
        torch.cuda.synchronize()
    finally:
        ops.set_exact_fp32(False)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / max(1e-12, float(b.double().abs().max())))
    assert rel(yg.detach().cpu(), y.detach()) < 3e-5
    assert rel(xg.grad.cpu(), xr.grad) < 5e-5
    assert rel(wg.grad.cpu(), wr.grad) < 5e-5

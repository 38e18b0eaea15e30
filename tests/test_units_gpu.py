"""GPU: the composable per-unit path (models/_unit.py) -- sub-modules of the mirrored R(2+1)D tree called on their own
(Conv3dBlock, SpatioTemporalConv, SpatioTemporalResBlock) and the fused head, against plain PyTorch-CPU modules with
the same parameters.  Tolerance 1e-4 of each tensor's scale (forward and gradients)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src.models import R2Plus1D as M

DEV = "cuda:0"


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(1e-9, float(b.double().abs().max())))


def ref_block(x, blk, training=True):
    """plain-torch evaluation of a mirrored Conv3dBlock (same parameter tensors, CPU)"""
    y = F.conv3d(x, blk.conv.weight, None, blk.conv.stride, blk.conv.padding)
    y = F.batch_norm(y, None, None, blk.bn.weight, blk.bn.bias, True, 0.1, 1e-5)
    return F.leaky_relu(y, blk.relu.negative_slope)


@pytest.mark.parametrize("cfg", [(8, 24, (1, 3, 3), (1, 2, 2), (0, 1, 1)), (24, 16, (3, 1, 1), (2, 1, 1), (1, 0, 0)),
                                 (5, 9, (3, 3, 3), (1, 1, 1), (1, 1, 1))])
def test_conv3dblock_standalone(cfg):
    cin, cout, k, s, p = cfg
    torch.manual_seed(1)
    blk = M.Conv3dBlock(cin, cout, k, s, 1, p, False, 0.1)
    with torch.no_grad():
        blk.bn.weight.uniform_(0.5, 1.5); blk.bn.bias.normal_(0, 0.2)
    x = torch.randn(3, cin, 6, 10, 9)
    xr = x.clone().requires_grad_(True)
    ref = ref_block(xr, blk)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    ref_gw, ref_gg, ref_gb = blk.conv.weight.grad.clone(), blk.bn.weight.grad.clone(), blk.bn.bias.grad.clone()
    blk.zero_grad()
    g = blk.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = g(xg)
    out.backward(dy.to(DEV))
    torch.cuda.synchronize()
    assert rel(out.cpu(), ref.detach()) < 1e-4
    assert rel(xg.grad.cpu(), xr.grad) < 1e-4
    assert rel(g.conv.weight.grad.cpu(), ref_gw) < 1e-4
    assert rel(g.bn.weight.grad.cpu(), ref_gg) < 1e-4 and rel(g.bn.bias.grad.cpu(), ref_gb) < 1e-4
    assert int(g.bn.num_batches_tracked) == 1


def test_resblock_standalone_matches_plain_torch():
    torch.manual_seed(2)
    blk = M.SpatioTemporalResBlock(8, 16, 3, downsample=True, alpha=0.2)
    x = torch.randn(2, 8, 6, 12, 12)

    def ref_forward(xx):
        def stc(m, v):
            return ref_block(ref_block(v, m.spatio_conv), m.temporal_conv)
        res = stc(blk.conv2, stc(blk.conv1, xx))
        return F.leaky_relu(stc(blk.downsample_conv, xx) + res, 0.2)

    ref = ref_forward(x)
    out = blk.to(DEV)(x.to(DEV))
    torch.cuda.synchronize()
    assert rel(out.cpu(), ref.detach()) < 1e-4


def test_eval_mode_unit_uses_running_statistics():
    torch.manual_seed(3)
    blk = M.Conv3dBlock(6, 10, (1, 3, 3), (1, 1, 1), 1, (0, 1, 1), False, 0.01)
    with torch.no_grad():
        blk.bn.running_mean.normal_(0, 0.3); blk.bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(2, 6, 3, 8, 8)
    y = F.conv3d(x, blk.conv.weight, None, blk.conv.stride, blk.conv.padding)
    ref = F.leaky_relu(F.batch_norm(y, blk.bn.running_mean.clone(), blk.bn.running_var.clone(), blk.bn.weight, blk.bn.bias,
                                    False, 0.1, 1e-5), 0.01)
    g = blk.to(DEV).eval()
    with torch.no_grad():
        out = g(x.to(DEV))
    torch.cuda.synchronize()
    assert rel(out.cpu(), ref.detach()) < 1e-4

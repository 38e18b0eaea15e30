"""Inference path: eval-mode forward of every BatchNorm-carrying model on the native kernels (running
statistics through md_bn_eval_params / the folded biases / the head kernels' eval branch, NoiseLayer and dropout off) against the
oracle restatements with training=False, on randomly moved running statistics.  The restatements are pinned in training mode by
the reference fixtures; eval mode is F.batch_norm(training=False) on the same code."""
import numpy as np
import pytest
import torch

from oracle import cnnlstm as oc
from oracle import mlstm_fcn as om
from oracle import r2plus1d as orc
from oracle import slowfast as osf
from oracle import transformer0d as ot

pytestmark = pytest.mark.gpu


def _randomise_running_stats(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if k.endswith("running_mean"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.3)
            elif k.endswith("running_var"):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif v.dim() == 1 and v.dtype.is_floating_point and ("bn" in k or "norm" in k or k.split(".")[-2] in ("1", "2")) and k.endswith("weight"):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)


def _check(m, x, ref_fn, tol=1e-3):
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = ref_fn(x, sd)
    m.cuda().eval()
    with torch.no_grad():
        out = m(x.cuda()).cpu()
        again = m(x.cuda()).cpu()
    assert float((out - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max())), float((out - ref).abs().max())
    assert torch.equal(out, again)                               # eval is deterministic and leaves the statistics alone
    after = m.state_dict()
    for k, v in sd.items():
        if "running" in k:
            assert torch.equal(after[k].cpu(), v), k


def test_slowfast_eval():
    from src.models.slowfast import SlowFast
    torch.manual_seed(1)
    m = SlowFast(input_shape=(3, 8, 64, 64), layers=[1, 1, 1, 1], alpha=4, tau_fast=1, num_classes=2, alpha_elu=1.0)
    _randomise_running_stats(m, 2)
    _check(m, torch.randn(3, 3, 8, 64, 64), lambda x, sd: osf.slowfast_forward(x, sd, [1, 1, 1, 1], 4, 1, 1.0, False))


def test_r2plus1d_eval():
    from src.models.R2Plus1D import R2Plus1DClassifier
    torch.manual_seed(3)
    m = R2Plus1DClassifier(input_size=(3, 5, 24, 24), num_classes=2, layer_sizes=[1, 1, 1, 1], alpha=0.01)
    _randomise_running_stats(m, 4)
    _check(m, torch.randn(3, 3, 5, 24, 24), lambda x, sd: orc.classifier_forward(x, sd, sd, [1, 1, 1, 1], 0.01, False))


def test_mlstm_fcn_eval():
    from src.models.MLSTM_FCN import MLSTM_FCN
    torch.manual_seed(5)
    m = MLSTM_FCN(n_features=14, fcn_dim=32, kernel_size=3, stride=1, seq_len=21, lstm_dim=24, lstm_n_layers=2, lstm_bidirectional=True,
                  lstm_dropout=0.2, reduction=16, alpha=0.01, n_classes=2)
    _randomise_running_stats(m, 6)
    _check(m, torch.randn(6, 21, 14), lambda x, sd: om.mlstm_fcn_forward(x, sd, 3, 1, 2, True, 0.01, False))


def test_cnnlstm_eval():
    from src.models.CnnLSTM import CnnLSTM
    torch.manual_seed(7)
    m = CnnLSTM(seq_len=21, n_features=12, conv_dim=32, conv_kernel=3, conv_stride=1, conv_padding=1, lstm_dim=32, n_layers=2,
                bidirectional=True, n_classes=2)
    _randomise_running_stats(m, 8)
    _check(m, torch.randn(6, 21, 12), lambda x, sd: oc.cnnlstm_forward(x, sd, 32, 2, True, False))


def test_transformer0d_eval():
    from src.models.transformer import Transformer
    torch.manual_seed(9)
    m = Transformer(n_features=18, kernel_size=5, feature_dims=64, max_len=21, n_layers=2, n_heads=4, dim_feedforward=96, dropout=0.3,
                    cls_dims=32, n_classes=2)
    _randomise_running_stats(m, 10)
    _check(m, torch.randn(8, 21, 18), lambda x, sd: ot.transformer0d_forward(x, sd, 2, 4, 5, False))

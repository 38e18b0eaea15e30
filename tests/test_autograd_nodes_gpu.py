"""GPU: the trunk's autograd node owns its activations.  Two training forwards of the same model before any
backward -- two clips, one summed loss, as a contrastive / two-view loop would do -- must give the same gradients as the sum
of two separate forward+backward passes; an eval-mode forward with grad enabled must refuse to backpropagate instead of
running the batch-statistics backward on a workspace that was never prepared for it."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from src.loss import FocalLoss
    from src.models.R2Plus1D import R2Plus1DClassifier

from oracle import r2plus1d as orc

DEV = "cuda:0"


def _model(seed=3):
    layers, alpha = [1, 1, 1, 1], 0.01
    m = R2Plus1DClassifier(input_size=(3, 5, 32, 32), num_classes=2, layer_sizes=layers, alpha=alpha)
    params, bufs = orc.synth_state(layers, seed, alpha)
    sd = dict(params); sd.update(bufs)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).train()


def test_two_forwards_then_backward_equals_two_separate_passes():
    x1 = orc.synth_clip(2, 5, 32, 11).to(DEV); x2 = orc.synth_clip(2, 5, 32, 12).to(DEV)
    y = orc.synth_labels(2, 11).to(DEV)
    loss_fn = FocalLoss(weight=torch.ones(2), gamma=2.0)
    m = _model()
    l1 = loss_fn(m(x1), y); l2 = loss_fn(m(x2), y)          # second forward BEFORE the first backward
    with torch.no_grad():
        m(x1 * 0.5)                                          # an interleaved probe forward must not disturb the saved activations
    (l1 + l2).backward()
    torch.cuda.synchronize()
    got = {k: p.grad.clone() for k, p in m.named_parameters()}
    ref = _model()
    loss_fn(ref(x1), y).backward(); loss_fn(ref(x2), y).backward()      # gradients accumulate
    torch.cuda.synchronize()
    for k, p in ref.named_parameters():
        assert torch.equal(got[k], p.grad), k               # same kernels on the same data: bit-identical


def test_backward_through_an_eval_forward_is_refused():
    m = _model().eval()
    x = orc.synth_clip(2, 5, 32, 5).to(DEV)
    out = m(x)
    with pytest.raises(RuntimeError):
        out.sum().backward()

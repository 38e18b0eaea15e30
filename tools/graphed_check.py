"""Whole-step HIP graph (src/utils/graphed.py) against the eager step: bit-identical loss and gradients for SlowFast (tiny) and
ViViT (dropout 0), a replay with a new batch, and ViViT with dropout running.  The eager reference steps run on a side stream and
drop their autograd graphs (src/utils/graphed.py explains why no earlier graph may outlive into a capture).
    python tools/graphed_check.py        -> prints "graphed_check OK" """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
from src.loss import FocalLoss, LDAMLoss
from src.models.MLSTM_FCN import MLSTM_FCN
from src.models.slowfast import SlowFast
from src.models.ViViT import ViViT
from src.utils.graphed import GraphedStep

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())


def eager(model, loss_fn, x, y):
    with torch.cuda.stream(side):
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model(x), y)
        loss.backward()
    side.synchronize()
    return float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters()}


torch.manual_seed(0)
cases = [("slowfast", SlowFast(input_shape=(3, 8, 64, 64), layers=[1, 1, 1, 1], alpha=4, tau_fast=1, num_classes=2),
          LDAMLoss(cls_num_list=[100, 2000], max_m=0.5, s=1.0, weight=None), (2, 3, 8, 64, 64)),
         ("vivit", ViViT(image_size=32, patch_size=8, n_frames=5, n_classes=2, dim=32, depth=2, n_heads=2, pool="mean", in_channels=3,
                         d_head=16, dropout=0.0, embedd_dropout=0.0, scale_dim=2), FocalLoss(gamma=2.0), (2, 3, 5, 32, 32))]
mlstm = MLSTM_FCN(n_features=6, fcn_dim=8, kernel_size=3, stride=1, seq_len=8, lstm_dim=64, lstm_n_layers=2, lstm_bidirectional=True,
                  lstm_dropout=0.0, reduction=4, alpha=0.01, n_classes=2)
mlstm.noise.std = 0.5                                # the CPU-generator noise must be the same stream eager and replayed
cases.append(("mlstm_fcn+noise", mlstm, FocalLoss(gamma=2.0), (2, 8, 6)))
for name, model, loss_fn, shape in cases:
    model = model.cuda().train()
    xs = [torch.randn(*shape, device="cuda") for _ in range(2)]
    ys = [torch.tensor([0, 1], device="cuda"), torch.tensor([1, 1], device="cuda")]
    state = {k: v.clone() for k, v in model.state_dict().items()}
    ref = []
    torch.manual_seed(77)                              # CPU generator: one NoiseLayer draw per step, eager and replayed alike
    for x, y in zip(xs, ys):                       # BatchNorm running statistics move: replay the same two-batch sequence
        ref.append(eager(model, loss_fn, x, y))
    model.load_state_dict(state)
    gs = GraphedStep(model, loss_fn, [xs[0]], ys[0], warmup=2)
    model.load_state_dict(state)                   # the warm-up and the capture pass moved the running statistics
    torch.manual_seed(77)
    for (rl, rg), x, y in zip(ref, xs, ys):
        _, loss = gs([x], y)
        torch.cuda.synchronize()
        assert float(loss.detach()) == rl, (name, float(loss.detach()), rl)
        bad = [k for k, p in model.named_parameters() if not torch.equal(p.grad, rg[k])]
        assert not bad, (name, bad[:4])
    print(name, "graph == eager on two batches")
m = ViViT(image_size=32, patch_size=8, n_frames=5, n_classes=2, dim=32, depth=1, n_heads=2, pool="cls", in_channels=3, d_head=16,
          dropout=0.3, embedd_dropout=0.1, scale_dim=2).cuda().train()
x = torch.randn(2, 3, 5, 32, 32, device="cuda"); y = torch.tensor([0, 1], device="cuda")
gs = GraphedStep(m, FocalLoss(gamma=2.0), [x], y, warmup=2)
losses = []
for _ in range(3):
    _, loss = gs([x], y); torch.cuda.synchronize(); losses.append(float(loss.detach()))
assert all(l == l for l in losses) and len(set(losses)) > 1, losses      # finite, and the dropout masks differ between replays
print("vivit with dropout: replays draw new masks", losses)
print("graphed_check OK")

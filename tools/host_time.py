"""Is the training step host-bound?  Time for the host to QUEUE n steps vs time for the GPU to finish them:
    python tools/host_time.py"""
import sys, os, time; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
from src.models.R2Plus1D import R2Plus1DClassifier
from src.loss import FocalLoss
from src.optim import ClipAdamW
dev = torch.device('cuda:0')
torch.manual_seed(1234)
model = R2Plus1DClassifier(input_size=(3, 21, 128, 128), num_classes=2, layer_sizes=[1, 2, 2, 1], alpha=0.01).to(dev).train()
loss_fn = FocalLoss(weight=torch.ones(2), gamma=2.0)
opt = ClipAdamW(model.parameters(), lr=2e-4)
x = torch.randn(8, 3, 21, 128, 128, device=dev) * 50; y = torch.tensor([0, 1, 0, 0, 1, 0, 0, 0], device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss_fn(model(x), y).backward()
    opt.step(max_norm=1.0)
for _ in range(5): step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host queues a step in {(t1-t0)/n*1e3:.3f} ms; GPU finishes a step in {(t2-t0)/n*1e3:.3f} ms")
# host cost with an idle GPU (sync after every step): pure launch + python time
ts = []
for _ in range(10):
    torch.cuda.synchronize(); a = time.perf_counter(); step(); ts.append(time.perf_counter() - a)
print(f"host-only step time (GPU idle at entry): {min(ts)*1e3:.3f} ms")
# ---- per-phase host time (no device sync inside; GPU drained before each step)
import collections
acc = collections.defaultdict(float)
for _ in range(10):
    torch.cuda.synchronize()
    a = time.perf_counter(); opt.zero_grad(set_to_none=True); b = time.perf_counter(); acc['zero_grad'] += b - a
    out = model(x); c = time.perf_counter(); acc['forward'] += c - b
    l = loss_fn(out, y); d = time.perf_counter(); acc['loss'] += d - c
    l.backward(); e = time.perf_counter(); acc['backward'] += e - d
    opt.step(max_norm=1.0); f = time.perf_counter(); acc['opt.step'] += f - e
print({k: round(v / 10 * 1e3, 3) for k, v in acc.items()})

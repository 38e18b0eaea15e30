"""Host time per phase of the data-parallel step on a ONE-rank RCCL group (no device synchronisation inside the loop):
    python tools/dp_host_time.py"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch, torch.distributed as dist
torch.set_num_threads(16)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from src.models.R2Plus1D import R2Plus1DClassifier
from src.loss import FocalLoss
from src.optim import ClipAdamW
from src.distributed import GradAllReducer
torch.manual_seed(1234)
model = R2Plus1DClassifier(input_size=(3, 21, 128, 128), num_classes=2, layer_sizes=[1, 2, 2, 1], alpha=0.01).to(dev).train()
red = GradAllReducer(model)
loss_fn = FocalLoss(weight=torch.ones(2), gamma=2.0)
opt = ClipAdamW(model.parameters(), lr=2e-4)
x = torch.randn(8, 3, 21, 128, 128, device=dev) * 50; y = torch.tensor([0, 1, 0, 0, 1, 0, 0, 0], device=dev)
acc = collections.defaultdict(float)
def step(rec):
    t = [time.perf_counter()]
    red.zero_grad(); t.append(time.perf_counter())
    out = model(x); t.append(time.perf_counter())
    loss = loss_fn(out, y); fin = torch.isfinite(loss.detach()).float(); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    ok = red.reduce_rest(fin); t.append(time.perf_counter())
    opt.step(max_norm=1.0, ok=ok); t.append(time.perf_counter())
    if rec:
        for k, a, b in zip(("zero_grad", "forward", "loss", "backward(+stage all-reduces)", "reduce_rest", "opt.step"), t, t[1:]):
            acc[k] += b - a
for _ in range(5): step(False)
torch.cuda.synchronize()
n = 20; t0 = time.perf_counter()
for _ in range(n): step(True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host queues a DP step in {(t1-t0)/n*1e3:.3f} ms; GPU finishes in {(t2-t0)/n*1e3:.3f} ms")
print({k: round(v / n * 1e3, 3) for k, v in acc.items()})
# the same with the GPU drained before every step: pure host cost
acc.clear()
for _ in range(10):
    torch.cuda.synchronize(); step(True)
print("GPU idle at entry:", {k: round(v / 10 * 1e3, 3) for k, v in acc.items()})
dist.destroy_process_group()

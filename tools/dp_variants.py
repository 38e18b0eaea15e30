"""Where does the data-parallel step's extra time go?  One-rank RCCL group, GPU time per step for variants of the step:
    python tools/dp_variants.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch, torch.distributed as dist
torch.set_num_threads(16)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from src.models.R2Plus1D import R2Plus1DClassifier
from src.loss import FocalLoss
from src.optim import ClipAdamW
from src.distributed import GradAllReducer, dp_train_step
torch.manual_seed(1234)
model = R2Plus1DClassifier(input_size=(3, 21, 128, 128), num_classes=2, layer_sizes=[1, 2, 2, 1], alpha=0.01).to(dev).train()
loss_fn = FocalLoss(weight=torch.ones(2), gamma=2.0)
opt = ClipAdamW(model.parameters(), lr=2e-4)
x = torch.randn(8, 3, 21, 128, 128, device=dev) * 50; y = torch.tensor([0, 1, 0, 0, 1, 0, 0, 0], device=dev)
def timeit(step, n=30):
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def plain():
    opt.zero_grad(set_to_none=True); loss_fn(model(x), y).backward(); opt.step(max_norm=1.0)
print(f"single-process step                      {timeit(plain):.3f} ms")
red = GradAllReducer(model)
def dp(): dp_train_step(model, red, opt, loss_fn, x, y, max_norm_grad=1.0)
print(f"data-parallel step                       {timeit(dp):.3f} ms")
real_avg = red._avg
red._avg = lambda t, async_op: None
print(f"  ... without the collectives            {timeit(dp):.3f} ms")
red._avg = real_avg
hook = red.trunk.grad_segment_hook; red.trunk.grad_segment_hook = None
def dp_nohook():
    red.zero_grad(); out = model(x); loss = loss_fn(out, y); fin = torch.isfinite(loss.detach()).float(); loss.backward()
    ok = red.reduce_rest(fin); opt.step(max_norm=1.0, ok=ok)
print(f"  ... one backward call, flag bucket only {timeit(dp_nohook):.3f} ms")
red.trunk.grad_segment_hook = hook
# stage hooks kept (stage-wise backward), but every collective issued after the last stage
calls = []
def late_hook(st, flat, grads, stream=None):
    calls.append((st, flat, grads))
    if st == 0:
        for a in calls:
            hook(a[0], a[1], a[2], None)
        calls.clear()
red.trunk.grad_segment_hook = late_hook
print(f"  ... all collectives after the last stage {timeit(dp):.3f} ms")
red.trunk.grad_segment_hook = hook
dist.destroy_process_group()

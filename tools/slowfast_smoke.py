"""SlowFast at the BASELINE cfg5 shape (T=32, 224x224, alpha=4, layers [1,2,2,1]) on the native kernels: one training step,
finite outputs/gradients, and a rough clips/s (correctness-first path: one layout conversion per unit).
    python tools/slowfast_smoke.py [B]"""
import sys, os, time; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
torch.set_num_threads(min(16, os.cpu_count() or 16))   # stay inside the box's CPU quota (see bench.py)
from src.models.slowfast import SlowFast
from src.loss import LDAMLoss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = SlowFast(input_shape=(3, 32, 224, 224), layers=[1, 2, 2, 1], alpha=4, tau_fast=1, num_classes=2).to(dev).train()
loss_fn = LDAMLoss(cls_num_list=[100, 2000], max_m=0.5, s=1.0, weight=None)
x = torch.randn(B, 3, 32, 224, 224, device=dev) * 50
y = (torch.arange(B) % 2).to(dev)
def step():
    m.zero_grad(set_to_none=True)
    out = m(x); loss = loss_fn(out, y); loss.backward(); return out, loss
out, loss = step(); torch.cuda.synchronize()
assert torch.isfinite(out).all() and torch.isfinite(loss)
bad = [k for k, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
assert not bad, bad[:5]
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"SlowFast B={B}: loss {float(loss):.4f}, {len(list(m.parameters()))} parameter tensors all with finite gradients, "
      f"{dt*1e3:.1f} ms/step = {B/dt:.1f} clips/s, peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
if os.environ.get("SF_PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); print(f"steady: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step")
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=10, max_name_column_width=60))
if os.environ.get("SF_GRAPH"):
    from src.utils.graphed import GraphedStep
    out, loss = step(); torch.cuda.synchronize()
    eager = {k: p.grad.clone() for k, p in m.named_parameters()}; eager_loss = float(loss.detach())
    del out, loss                                  # no autograd graph of an eager step may outlive into the capture (src/utils/graphed.py)
    torch.cuda.synchronize()
    gs = GraphedStep(m, loss_fn, [x], y)
    o2, l2 = gs([x], y); torch.cuda.synchronize()
    same = all(torch.equal(eager[k], p.grad) for k, p in m.named_parameters())
    print("graph vs eager: loss", float(l2.detach()), eager_loss, "grads identical:", same, flush=True)
    for _ in range(3): gs([x], y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): gs([x], y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"graphed: {dt*1e3:.2f} ms/step = {B/dt:.1f} clips/s")

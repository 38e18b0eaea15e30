"""ViViT at the BASELINE cfg3 shape (B=4, T=21, 224x224, patch 16, dim 128, depth 2, heads 4, d_head 64, scale_dim 8, pool mean)
on the composable native path: one training step (forward, Focal loss, backward), timed after warm-up.  Run on the GPU box:
python tools/vivit_smoke.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
torch.set_num_threads(min(16, os.cpu_count() or 16))   # stay inside the box's CPU quota (see bench.py)
from src.models.ViViT import ViViT
from src.loss import FocalLoss

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
m = ViViT(image_size=224, patch_size=16, n_frames=21, n_classes=2, dim=128, depth=2, n_heads=4, pool="mean", in_channels=3, d_head=64,
          dropout=0.1, embedd_dropout=0.1, scale_dim=8).cuda().train()
loss_fn = FocalLoss(gamma=2.0)
x = torch.randn(4, 3, 21, 224, 224, device="cuda"); y = torch.randint(0, 2, (4,), device="cuda")


def step():
    for p in m.parameters():
        p.grad = None
    loss = loss_fn(m(x), y)
    loss.backward()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"ViViT cfg3 B=4: {dt * 1e3:.2f} ms/step  ({4 / dt:.1f} clips/s)  loss {float(loss.detach()):.4f}")
import json
print(json.dumps({"metric": "clips/sec (fwd+bwd) ViViT cfg3", "value": round(4 / dt, 1), "unit": "clips/s", "n_gpus": 1, "steps": steps,
                  "warmup": 3, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "dtype": "f32 (attention: fp32 MFMA; Linears: 3 fp16/bf16 MFMAs on hi+lo splits)",
                  "data": "synthetic", "alg_tflops": round(4 / dt * 27.1e9 / 1e12, 2),
                  "config": {"workload": "BASELINE configs[2]: ViViT (B=4,3,21,224,224), patch 16, dim 128, depth 2, heads 4, d_head 64, scale_dim 8, pool mean, dropout 0.1; forward + FocalLoss + backward (no optimizer step)"}}))
if os.environ.get("VIVIT_PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))

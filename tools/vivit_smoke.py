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


def attention_view():
    """Time of the matrix-core attention kernels in one step (torch profiler) against their algorithmic FLOPs: per (sequence,
    head) and layer 12 S^2 d (QK^T, PV forward; dP, dq, dk, dv backward), space S = 197 over B*T sequences, temporal S = 22 over B."""
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step(); torch.cuda.synchronize()
    us = sum(e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total for e in prof.key_averages() if "k_attn_" in e.key)
    fl = 2 * 4 * 64 * 12.0 * (84 * 197 ** 2 + 4 * 22 ** 2)          # depth 2 x heads 4 x d_head 64
    return us, fl


def cpu_baseline(nsteps=2, warmup=1):
    """The oracle's ViViT (oracle/vivit.py, dropout off) forward + Focal loss + backward on the host cores, same shapes."""
    from oracle import losses as ol, vivit as ov
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    xc, yc = x.cpu(), y.cpu()
    one = torch.ones(2)
    def cstep():
        for v in sd.values():
            v.grad = None
        ol.focal_loss(ov.vivit_forward(xc, sd, 16, 2, 4, "mean", 3, 1.0, True), yc, one, 2.0).backward()
    for _ in range(warmup):
        cstep()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        cstep()
    return {"value": round(4 * nsteps / (time.perf_counter() - t0), 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{nsteps} forward+loss+backward steps (after {warmup} warm-up) of the same B=4 cfg3 workload, oracle/vivit.py on torch-CPU, dropout off"}


att_us, att_fl = attention_view()
alg = 4 / dt * 27.1e9 / 1e12
out = {"metric": "clips/sec (fwd+bwd) ViViT cfg3", "value": round(4 / dt, 1), "unit": "clips/s", "n_gpus": 1, "steps": steps,
       "warmup": 3, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True,
       "dtype": "f32 storage; attention and Linears: every product as 3 fp16 (forward) / bf16 (backward) MFMAs on hi+lo splits, f32 accumulate", "data": "synthetic", "alg_tflops": round(alg, 2),
       "config": {"workload": "BASELINE configs[2]: ViViT (B=4,3,21,224,224), patch 16, dim 128, depth 2, heads 4, d_head 64, scale_dim 8, pool mean, dropout 0.1; forward + FocalLoss + backward (no optimizer step); eager launch sequence (tools/vivit_graph.py: the same step as one HIP graph)"},
       "roofline": {"bound": "mfma", "achieved": round(alg, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(alg / 2500.0, 5), "traffic": None,
                    "kernel": "whole step: 27.1 GFLOP/clip algorithmic (fwd+bwd) / step time, against the dense 16-bit MFMA peak north_star names for this config",
                    "vs_fp32_matrix_peak_157": round(alg / 157.0, 4),
                    "attention": {"kernels": "k_attn_mfma_fwd<.., SP, LSE>, k_attn_lse_bwd_{q,kv}", "time_us_per_step": round(att_us, 1), "alg_gflop_per_step": round(att_fl / 1e9, 2),
                                  "achieved_tflops": round(att_fl / (att_us * 1e-6) / 1e12, 2) if att_us else None,
                                  "frac_of_split_peak_833": round(att_fl / (att_us * 1e-6) / 1e12 / (2500.0 / 3.0), 4) if att_us else None,
                                  "note": "algorithmic FLOPs / kernel time; 833 = dense 16-bit MFMA peak / 3 issued products per multiply (MD_ATTN_SPLIT=0: the exact fp32-MFMA kernels, peak 157)",
                                  "frac_of_2500": round(att_fl / (att_us * 1e-6) / 1e12 / 2500.0, 5) if att_us else None}}}
if not os.environ.get("NO_CPU_BASELINE"):
    out["cpu_baseline"] = cpu_baseline()
print(json.dumps(out))
if os.environ.get("VIVIT_PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))

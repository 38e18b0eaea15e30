"""ViViT cfg3 (see tools/vivit_smoke.py) with the whole step (forward + Focal loss + backward) replayed as one HIP graph
(src/utils/graphed.py).  Fresh process, nothing eager on the default stream.   python tools/vivit_graph.py [steps] [--full]     (--full: roofline object incl. the attention kernels, and the CPU baseline)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
torch.set_num_threads(min(16, os.cpu_count() or 16))   # stay inside the box's CPU quota (see bench.py)
from src.loss import FocalLoss
from src.models.ViViT import ViViT
from src.utils.graphed import GraphedStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50
torch.manual_seed(0)
m = ViViT(image_size=224, patch_size=16, n_frames=21, n_classes=2, dim=128, depth=2, n_heads=4, pool="mean", in_channels=3, d_head=64,
          dropout=float(os.environ.get("VIVIT_DROPOUT", "0.1")), embedd_dropout=float(os.environ.get("VIVIT_DROPOUT", "0.1")), scale_dim=8).cuda().train()
x = torch.randn(4, 3, 21, 224, 224, device="cuda"); y = torch.randint(0, 2, (4,), device="cuda")
gs = GraphedStep(m, FocalLoss(gamma=2.0), [x], y)
for _ in range(3):
    gs([x], y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    _, loss = gs([x], y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
def attention_view():
    """Time of the matrix-core attention kernels in ONE eager step (torch profiler; the kernels are the ones the graph replays) against
    their algorithmic FLOPs: per (sequence, head) and layer 12 S^2 d (QK^T, PV forward; dP, dq, dk, dv backward), space S = 197 over
    B*T sequences, temporal S = 22 over B."""
    from torch.profiler import profile, ProfilerActivity
    lf = FocalLoss(gamma=2.0)
    def estep():
        for p in m.parameters():
            p.grad = None
        lf(m(x), y).backward()
    estep(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        estep(); torch.cuda.synchronize()
    us = sum(e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total for e in prof.key_averages() if "k_attn_" in e.key)
    fl = 2 * 4 * 64 * 12.0 * (84 * 197 ** 2 + 4 * 22 ** 2)          # depth 2 x heads 4 x d_head 64
    return us, fl


def cpu_baseline(nsteps=2, warmup=1):
    """The oracle's ViViT (oracle/vivit.py, dropout off) forward + Focal loss + backward on the host cores, same shapes (the checker
    timed as the reported CPU baseline, as in bench.py; nothing here runs through it)."""
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


alg = 4 / dt * 27.1e9 / 1e12
out = {"metric": "clips/sec (fwd+bwd) ViViT cfg3, whole step as one HIP graph", "value": round(4 / dt, 1), "unit": "clips/s",
       "n_gpus": 1, "steps": steps, "warmup": 3, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True,
       "dtype": "f32 storage; attention and Linears: every product as 3 fp16 (forward) / bf16 (backward) MFMAs on hi+lo splits, f32 accumulate",
       "data": "synthetic", "loss": float(loss.detach()), "alg_tflops": round(alg, 2),
       "config": {"workload": "BASELINE configs[2]: ViViT (B=4,3,21,224,224), patch 16, dim 128, depth 2, heads 4, d_head 64, scale_dim 8, pool mean, dropout 0.1; forward + FocalLoss + backward, GraphedStep"}}
if "--full" in sys.argv:
    att_us, att_fl = attention_view()
    att_tf = att_fl / (att_us * 1e-6) / 1e12 if att_us else None
    out["roofline"] = {"bound": "mfma", "achieved": round(alg, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(alg / 2500.0, 5), "traffic": None,
                       "kernel": "whole step: 27.1 GFLOP/clip algorithmic (fwd+bwd) / replayed step time, against the dense 16-bit MFMA peak north_star names for this config",
                       "frac_of_split_peak_833": round(alg / (2500.0 / 3.0), 5),
                       "attention": {"kernels": "k_attn_mfma_fwd<.., SP, LSE>, k_attn_lse_bwd_{q,kv}", "time_us_per_step": round(att_us, 1),
                                     "alg_gflop_per_step": round(att_fl / 1e9, 2), "achieved_tflops": round(att_tf, 2) if att_tf else None,
                                     "frac_of_split_peak_833": round(att_tf / (2500.0 / 3.0), 4) if att_tf else None,
                                     "frac_of_2500": round(att_tf / 2500.0, 5) if att_tf else None,
                                     "note": "algorithmic FLOPs / kernel time of one eager step (the same kernels the graph replays); 833 = dense 16-bit MFMA peak / 3 issued products per multiply"}}
    out["cpu_baseline"] = cpu_baseline()
print(json.dumps(out))

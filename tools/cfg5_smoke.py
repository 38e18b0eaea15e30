"""BASELINE config 5 on one GPU: SlowFast [1,2,2,1] (alpha 4, 3x32x224x224) + MLSTM_FCN (14 features x 21 steps, fcn 128, LSTM 128x4
bidirectional) fused by the MultiModalModel_GB recipe (src/models/fusion.py), GradientBlending(0.1/0.4/0.5) over LDAMLoss
(cls_num_list [100, 2000], max_m 0.5, s 1.0) with the DRW class weights of the last schedule quarter (beta 0.75), ClipAdamW step.
python tools/cfg5_smoke.py [B] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import numpy as np
import torch
torch.set_num_threads(min(16, os.cpu_count() or 16))   # stay inside the box's CPU quota (see bench.py)
from src.GradientBlending import GradientBlending
from src.loss import LDAMLoss
from src.models.fusion import FusionGB
from src.models.MLSTM_FCN import MLSTM_FCN
from src.models.slowfast import SlowFast
from src.optim import ClipAdamW

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(0)
vis = SlowFast(input_shape=(3, 32, 224, 224), layers=[1, 2, 2, 1], alpha=4, tau_fast=1, num_classes=2)
ts = MLSTM_FCN(n_features=14, fcn_dim=128, kernel_size=3, stride=1, seq_len=21, lstm_dim=128, lstm_n_layers=4, lstm_bidirectional=True,
               lstm_dropout=0.1, reduction=16, alpha=0.01, n_classes=2)
m = FusionGB(2, vis, ts).cuda().train()
cls_num = [100, 2000]
beta = 0.75                                                        # DRW, last quarter (src/train.py:318-329)
w = (1.0 - beta) / (1.0 - np.power(beta, cls_num)); w = w / w.sum() * len(cls_num)
loss_fn = LDAMLoss(cls_num, max_m=0.5, weight=torch.tensor(w, dtype=torch.float32).cuda(), s=1.0)
gb = GradientBlending(loss_fn, loss_fn, loss_fn, 0.1, 0.4, 0.5)
opt = ClipAdamW(m.parameters(), lr=2e-4, max_norm=1.0)
xv = torch.randn(B, 3, 32, 224, 224, device="cuda") * 50; xt = torch.randn(B, 21, 14, device="cuda")
y = (torch.arange(B) % 2).cuda()


def cpu_baseline(nsteps=2, warmup=1):
    """oracle/fusion.py (SlowFast + MLSTM_FCN restatements, dropout off) forward + blended LDAM loss + backward on the host cores."""
    from oracle import fusion as ofu, losses as ol
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    leaves = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    xv_c, xt_c, y_c = xv.cpu(), xt.cpu(), y.cpu()
    m_list = ol.ldam_margins(cls_num, 0.5)
    wc = torch.tensor(w, dtype=torch.float32)
    cfg = dict(kernel_size=3, stride=1, lstm_n_layers=4, bidirectional=True, alpha=0.01)
    def cstep():
        for v in leaves:
            v.grad = None
        o = ofu.slowfast_mlstm_forward(xv_c, xt_c, sd, [1, 2, 2, 1], 4, 1.0, cfg, True)
        l = [ol.ldam_loss(t, y_c, m_list, wc, 1.0) for t in o]
        ol.gradient_blending(l[0], l[1], l[2], 0.1, 0.4, 0.5).backward()
    for _ in range(warmup):
        cstep()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        cstep()
    return {"value": round(B * nsteps / (time.perf_counter() - t0), 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{nsteps} forward+loss+backward steps (after {warmup} warm-up) of the same B={B} cfg5 workload, oracle/fusion.py on torch-CPU, no optimizer step"}



if os.environ.get("CFG5_GRAPH"):
    # whole forward + loss + backward as one HIP graph (fresh process: nothing eager on the default stream before the capture)
    from src.utils.graphed import GraphedStep
    gs = GraphedStep(m, lambda a, b, c, t: gb(a, b, c, t), [xv, xt], y)
    def gstep():
        _, loss = gs([xv, xt], y)
        opt.step()
        return loss
    for _ in range(3):
        gstep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        loss = gstep()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    assert bool(torch.isfinite(loss))
    gbs = B / dt * 686e6 / 1e9
    out = {"metric": "clips/sec (full step) SlowFast + MLSTM_FCN, GradientBlending over LDAM + DRW weights; forward+loss+backward as one HIP graph",
           "value": round(B / dt, 1), "unit": "clips/s", "n_gpus": 1, "steps": steps, "warmup": 3, "ms_per_step": round(dt * 1e3, 3),
           "higher_is_better": True, "data": "synthetic", "loss": float(loss.detach()), "dtype": "f32 storage; 3 fp16/bf16 MFMAs per product",
           "config": {"workload": f"BASELINE configs[4] on ONE GPU: SlowFast [1,2,2,1] alpha 4 ({B},3,32,224,224) + MLSTM_FCN (14x21, fcn 128, LSTM 128x4 bi), FusionGB, GradientBlending(0.1/0.4/0.5) over LDAM(max_m 0.5, s 1) with DRW weights (beta 0.75); forward + loss + backward replayed by GraphedStep, ClipAdamW(2e-4, clip 1.0) eager"},
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "traffic": None,
                        "kernel": "whole step: clips/s x 686 MB/clip (SURVEY 8(d): SlowFast conv I/O, fp32 storage, fwd+bwd); a whole-job figure, the step is a dependent chain of ~1100 small kernels"}}
    if not os.environ.get("NO_CPU_BASELINE"):
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))
    sys.exit(0)


def step():
    opt.zero_grad(set_to_none=True)
    o = m(xv, xt)
    loss = gb(o[0], o[1], o[2], y)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
assert bool(torch.isfinite(loss))
gbs = B / dt * 686e6 / 1e9
out = {"metric": "clips/sec (full step) SlowFast + MLSTM_FCN, GradientBlending over LDAM + DRW weights", "value": round(B / dt, 1),
       "unit": "clips/s", "n_gpus": 1, "steps": steps, "warmup": 3, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True,
       "data": "synthetic", "loss": float(loss.detach()), "dtype": "f32 storage; 3 fp16/bf16 MFMAs per product",
       "config": {"workload": f"BASELINE configs[4] on ONE GPU: SlowFast [1,2,2,1] alpha 4 ({B},3,32,224,224) + MLSTM_FCN (14x21, fcn 128, LSTM 128x4 bi), FusionGB, GradientBlending(0.1/0.4/0.5) over LDAM(max_m 0.5, s 1) with DRW weights (beta 0.75), ClipAdamW(2e-4, clip 1.0)"},
       "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "traffic": None,
                    "kernel": "whole step: clips/s x 686 MB/clip (SURVEY 8(d): SlowFast conv I/O, fp32 storage, fwd+bwd) -- the composable path is host-bound, so this is a whole-job figure, not a kernel's"}}
if not os.environ.get("NO_CPU_BASELINE"):
    out["cpu_baseline"] = cpu_baseline()
print(json.dumps(out))
if os.environ.get("CFG5_PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))

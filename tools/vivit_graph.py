"""ViViT cfg3 (see tools/vivit_smoke.py) with the whole step (forward + Focal loss + backward) replayed as one HIP graph
(src/utils/graphed.py).  Fresh process, nothing eager on the default stream.   python tools/vivit_graph.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
torch.set_num_threads(min(16, os.cpu_count() or 16))   # stay inside the box's CPU quota (see bench.py)
from src.loss import FocalLoss
from src.models.ViViT import ViViT
from src.utils.graphed import GraphedStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.manual_seed(0)
m = ViViT(image_size=224, patch_size=16, n_frames=21, n_classes=2, dim=128, depth=2, n_heads=4, pool="mean", in_channels=3, d_head=64,
          dropout=0.1, embedd_dropout=0.1, scale_dim=8).cuda().train()
x = torch.randn(4, 3, 21, 224, 224, device="cuda"); y = torch.randint(0, 2, (4,), device="cuda")
gs = GraphedStep(m, FocalLoss(gamma=2.0), [x], y)
for _ in range(3):
    gs([x], y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    _, loss = gs([x], y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(json.dumps({"metric": "clips/sec (fwd+bwd) ViViT cfg3, whole step as one HIP graph", "value": round(4 / dt, 1), "unit": "clips/s",
                  "n_gpus": 1, "steps": steps, "warmup": 3, "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "data": "synthetic",
                  "loss": float(loss.detach()), "alg_tflops": round(4 / dt * 27.1e9 / 1e12, 2),
                  "config": {"workload": "BASELINE configs[2]: ViViT (B=4,3,21,224,224), patch 16, dim 128, depth 2, heads 4, d_head 64, scale_dim 8, pool mean, dropout 0.1; forward + FocalLoss + backward, GraphedStep"}}))

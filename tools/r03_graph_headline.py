"""Headline step (R(2+1)D B=8) forward + Focal loss + backward: eager from the executor vs replayed as one HIP graph (one-stream schedule).
python tools/r03_graph_headline.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
torch.set_num_threads(min(16, os.cpu_count() or 16))
from src.loss import FocalLoss
from src.models.R2Plus1D import R2Plus1DClassifier
from src.utils.graphed import GraphedStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.manual_seed(0)
m = R2Plus1DClassifier(input_size=(3, 21, 128, 128), num_classes=2, layer_sizes=[1, 2, 2, 1], alpha=0.01).cuda().train()
x = torch.randn(8, 3, 21, 128, 128, device="cuda"); y = torch.randint(0, 2, (8,), device="cuda")
lf = FocalLoss(weight=torch.ones(2), gamma=2.0)

def eager():
    for p in m.parameters(): p.grad = None
    loss = lf(m(x), y); loss.backward(); return loss
for _ in range(5): eager()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): eager()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / steps
gs = GraphedStep(m, lf, [x], y)
for _ in range(3): gs([x], y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): gs([x], y)
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / steps
print(json.dumps({"eager_ms": round(te * 1e3, 3), "graph_ms": round(tg * 1e3, 3)}))

"""One-shot probe of the capture crash: eager default-stream steps whose loss tensor stays alive, then a
capture.  Prints the Python stack on a fatal signal; run with AMD_LOG_LEVEL=3 to get the HIP API tail:
    AMD_LOG_LEVEL=3 python tools/graph_capture_probe.py [keep|drop] 2> hip.log"""
import faulthandler, sys, os; faulthandler.enable(all_threads=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
from src.models.slowfast import SlowFast
from src.loss import LDAMLoss
mode = sys.argv[1] if len(sys.argv) > 1 else "keep"
m = SlowFast(input_shape=(3, 8, 64, 64), layers=[1, 1, 1, 1], alpha=4, tau_fast=1, num_classes=2).cuda().train()
loss_fn = LDAMLoss(cls_num_list=[100, 2000], max_m=0.5, s=1.0, weight=None)
x = torch.randn(2, 3, 8, 64, 64, device="cuda"); y = torch.tensor([0, 1], device="cuda")
for _ in range(2):                      # eager, on the legacy default stream
    m.zero_grad(set_to_none=True); l = loss_fn(m(x), y); l.backward()
torch.cuda.synchronize()
if mode == "drop":
    del l
m.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
print("capturing; eager loss tensor alive:", mode == "keep", flush=True)
with torch.cuda.graph(g):
    out = m(x)
    l2 = loss_fn(out, y)
    l2.backward()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize(); print("replayed", float(out.sum()), flush=True)

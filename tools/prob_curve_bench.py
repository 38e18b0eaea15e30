"""Sliding-window inference throughput of the probability-curve path (src/utils/prob_curve.py) for the headline model shape:
R2Plus1D [1,2,2,1], windows of 21 frames at 128x128 cut from a (F,256,256,3) uint8 frame stack resident in HBM.
    python tools/prob_curve_bench.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
torch.set_num_threads(16)
from src.models.R2Plus1D import R2Plus1DClassifier
from src.utils.prob_curve import video_window_probabilities
torch.manual_seed(0)
if len(sys.argv) > 1 and sys.argv[1] == "slowfast":     # SlowFast [1,2,2,1], windows of 32 frames at 224x224 (set MD_GRAPH_STEP=1 for the graph replay)
    from src.models.slowfast import SlowFast
    m = SlowFast(input_shape=(3, 32, 224, 224), layers=[1, 2, 2, 1], alpha=4, tau_fast=1, num_classes=2).cuda().eval()
    T, CROP, F = 32, 224, 32 + 3 + 128
else:
    m = R2Plus1DClassifier(input_size=(3, 21, 128, 128), num_classes=2, layer_sizes=[1, 2, 2, 1], alpha=0.01).cuda().eval()
    T, CROP, F = 21, 128, 21 + 3 + 256
frames = torch.randint(0, 256, (F, 256, 256, 3), dtype=torch.uint8, device="cuda")
out = {}
for w in (1, 4, 16):
    video_window_probabilities(m, frames, T, 3, 0, F, CROP, w)          # warm-up (plans for this batch size)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p, c = video_window_probabilities(m, frames, T, 3, 0, F, CROP, w)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out["windows_per_launch_%d" % w] = {"windows": len(p), "windows_per_s": round(len(p) / dt, 1), "ms_per_window": round(dt / len(p) * 1e3, 3)}
print(json.dumps({"metric": "sliding-window inference, %s from a uint8 frame stack in HBM (eval mode, softmax + arg-max on the device, one read-back per shot)" % ("SlowFast T=32 224x224" if T == 32 else "R2Plus1D T=21 128x128"),
                  "graph_replay": os.environ.get("MD_GRAPH_STEP") == "1", **out}))

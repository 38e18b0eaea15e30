"""Batch-1 forward latencies of the six models the reference times in its notebook with `measure_computation_time`
(src/utils/utility.py:1201-1230: eval mode, no_grad, zeros input created on the host, the host->device copy inside the timed
region, 16 samples) - re-measured here WITH a device synchronisation inside the timed region (the reference's figures have none,
BASELINE.md section 1), after 3 warm-up calls.  Context numbers (inference, not the bench metric).   python tools/latency.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import numpy as np
import torch
torch.set_num_threads(min(16, os.cpu_count() or 16))   # stay inside the box's CPU quota (see bench.py)
from src.models.CnnLSTM import CnnLSTM
from src.models.MLSTM_FCN import MLSTM_FCN
from src.models.R2Plus1D import R2Plus1DClassifier
from src.models.slowfast import SlowFast
from src.models.transformer import Transformer
from src.models.ViViT import ViViT

torch.manual_seed(0)
CASES = [
    ("R2Plus1D [1,2,2,1] alpha 0.01", lambda: R2Plus1DClassifier(input_size=(3, 21, 128, 128), num_classes=2, layer_sizes=[1, 2, 2, 1], alpha=0.01),
     (1, 3, 21, 128, 128), 119.8),
    ("SlowFast [1,2,2,1] alpha 4", lambda: SlowFast(input_shape=(3, 20, 128, 128), layers=[1, 2, 2, 1], alpha=4, tau_fast=1, num_classes=2),
     (1, 3, 20, 128, 128), 29.0),
    ("ViViT 128, patch 16, dim 128, depth 4, heads 8", lambda: ViViT(image_size=128, patch_size=16, n_frames=21, n_classes=2, dim=128, depth=4, n_heads=8),
     (1, 3, 21, 128, 128), 34.1),
    ("Transformer-0D d 128, L4, H8, FF 512", lambda: Transformer(n_features=12, kernel_size=5, feature_dims=128, max_len=21, n_layers=4, n_heads=8,
                                                                 dim_feedforward=512, dropout=0.1, cls_dims=64, n_classes=2), (1, 21, 12), 8.46),
    ("CnnLSTM conv 32, LSTM 64x2", lambda: CnnLSTM(seq_len=21, n_features=12, conv_dim=32, conv_kernel=3, conv_stride=1, conv_padding=1,
                                                   lstm_dim=64, n_layers=2, bidirectional=True, n_classes=2), (1, 21, 12), 16.6),
    ("MLSTM_FCN fcn 64, LSTM 64x2", lambda: MLSTM_FCN(n_features=12, fcn_dim=64, kernel_size=3, stride=1, seq_len=21, lstm_dim=64, lstm_n_layers=2,
                                                     lstm_bidirectional=True, lstm_dropout=0.1, reduction=16, alpha=0.01, n_classes=2), (1, 21, 12), 18.7),
]
rows = []
import gc
for name, make, shape, published in CASES:
    m = make().cuda().eval()
    if os.environ.get('LAT_GC') == 'freeze':
        gc.collect(); gc.freeze()
    elif os.environ.get('LAT_GC') == 'off':
        gc.disable()
    ts = []
    with torch.no_grad():
        for i in range(3 + 16):
            x = torch.zeros(shape)
            t0 = time.time()
            out = m(x.cuda())
            torch.cuda.synchronize()
            if i >= 3:
                ts.append(time.time() - t0)
    rows.append({"model": name, "input": list(shape), "ms_mean": round(float(np.mean(ts)) * 1e3, 3), "ms_std": round(float(np.std(ts)) * 1e3, 3),
                 "ms_median": round(float(np.median(ts)) * 1e3, 3), "ms_samples": [round(t * 1e3, 2) for t in ts],
                 "reference_rtx3090_unsynchronised_ms": published})
    print(f"{name:50s} median {rows[-1]['ms_median']:8.3f} ms, mean {rows[-1]['ms_mean']:8.3f} +- {rows[-1]['ms_std']:.3f}   (reference notebook, RTX 3090, no sync: {published} ms)")
print(json.dumps({"metric": "batch-1 forward latency, eval mode, host->device copy and device sync inside the timed region", "unit": "ms",
                  "higher_is_better": False, "n_gpus": 1, "samples": 16, "warmup": 3, "rows": rows}))

"""src.train.train_per_epoch on the cfg5 pair (SlowFast [1,2,2,1] 32x224x224 + MLSTM_FCN, GradientBlending over LDAM, ClipAdamW), B=4,
batches already on the GPU: seconds per epoch of NB batches, eager and with MD_GRAPH_STEP=1 (forward+loss+backward replayed from a
HIP graph).   python tools/loop_time.py [NB]      (set MD_GRAPH_STEP=1 for the replayed loop)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
torch.set_num_threads(min(16, os.cpu_count() or 16))
import src.train as tr
from src.GradientBlending import GradientBlending
from src.loss import LDAMLoss
from src.models.fusion import FusionGB
from src.models.MLSTM_FCN import MLSTM_FCN
from src.models.slowfast import SlowFast
from src.optim import ClipAdamW

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = 4
torch.manual_seed(0)
vis = SlowFast(input_shape=(3, 32, 224, 224), layers=[1, 2, 2, 1], alpha=4, tau_fast=1, num_classes=2)
ts = MLSTM_FCN(n_features=14, fcn_dim=128, kernel_size=3, stride=1, seq_len=21, lstm_dim=128, lstm_n_layers=4, lstm_bidirectional=True,
               lstm_dropout=0.1, reduction=16, alpha=0.01, n_classes=2)
m = FusionGB(2, vis, ts).cuda()
ld = LDAMLoss([100, 2000], max_m=0.5, weight=torch.tensor([1.0, 1.0]).cuda(), s=1.0)
gb = GradientBlending(ld, ld, ld, 0.1, 0.4, 0.5)
opt = ClipAdamW(m.parameters(), lr=2e-4, max_norm=1.0)
batches = [({"video": torch.randn(B, 3, 32, 224, 224, device="cuda") * 50, "0D": torch.randn(B, 21, 14, device="cuda")},
            (torch.arange(B) % 2).cuda()) for _ in range(4)]
loader = [batches[i % 4] for i in range(NB)]
tr.train_per_epoch(loader[:6], m, opt, None, gb, "cuda:0", 1.0, "multi-GB")          # warm-up (and the capture, in graph mode)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = tr.train_per_epoch(loader, m, opt, None, gb, "cuda:0", 1.0, "multi-GB")
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"metric": "clips/sec, src.train.train_per_epoch (cfg5 pair, B=4)", "graph_step": tr._GRAPH_STEPS, "value": round(NB * B / dt, 1),
                  "unit": "clips/s", "ms_per_batch": round(dt / NB * 1e3, 3), "batches": NB, "loss": res[0]}))

"""One training step of each 0D model at the reference script's default hyper-parameters and batch size (train_0D_network.py:70-135:
batch 256, 21 steps, 12 features) and of the vision models at larger batches: catches size limits of the small fused kernels.
python tools/defaults_sweep.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
torch.set_num_threads(min(16, os.cpu_count() or 16))
from src.loss import FocalLoss
from src.models.CnnLSTM import CnnLSTM
from src.models.MLSTM_FCN import MLSTM_FCN
from src.models.R2Plus1D import R2Plus1DClassifier
from src.models.slowfast import SlowFast
from src.models.transformer import Transformer
from src.models.ViViT import ViViT

torch.manual_seed(0)
F = 12
CASES = [
    ("Transformer B=256", lambda: Transformer(n_features=F, kernel_size=5, feature_dims=128, max_len=21, n_layers=4, n_heads=8, dim_feedforward=1024,
                                              dropout=0.1, cls_dims=128, n_classes=2), (256, 21, F)),
    ("CnnLSTM B=256", lambda: CnnLSTM(seq_len=21, n_features=F, conv_dim=64, conv_kernel=3, conv_stride=1, conv_padding=1, lstm_dim=128,
                                      n_layers=4, bidirectional=True, n_classes=2), (256, 21, F)),
    ("MLSTM_FCN B=256", lambda: MLSTM_FCN(n_features=F, fcn_dim=128, kernel_size=3, stride=1, seq_len=21, lstm_dim=128, lstm_n_layers=4,
                                          lstm_bidirectional=True, lstm_dropout=0.1, reduction=16, alpha=0.01, n_classes=2), (256, 21, F)),
    ("R2Plus1D B=32", lambda: R2Plus1DClassifier(input_size=(3, 21, 128, 128), num_classes=2, layer_sizes=[1, 2, 2, 1], alpha=0.01), (32, 3, 21, 128, 128)),
    ("SlowFast B=32 (128x128, T=20)", lambda: SlowFast(input_shape=(3, 20, 128, 128), layers=[1, 2, 2, 1], alpha=4, tau_fast=1, num_classes=2),
     (32, 3, 20, 128, 128)),
    ("ViViT B=16 (128x128)", lambda: ViViT(image_size=128, patch_size=16, n_frames=21, n_classes=2, dim=128, depth=4, n_heads=8, dropout=0.1),
     (16, 3, 21, 128, 128)),
]
loss_fn = FocalLoss(torch.tensor([1.0, 1.0]).cuda(), 2.0)
bad = 0
for name, make, shape in CASES:
    try:
        m = make().cuda().train()
        x = torch.randn(*shape, device="cuda"); y = (torch.arange(shape[0], device="cuda") % 2)
        for _ in range(2):
            m.zero_grad(set_to_none=True)
            loss = loss_fn(m(x), y); loss.backward()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            m.zero_grad(set_to_none=True)
            loss = loss_fn(m(x), y); loss.backward()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        ok = bool(torch.isfinite(loss)) and all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
        print(f"{name:34s} {'ok' if ok else 'NON-FINITE'}  {dt * 1e3:8.2f} ms/step  loss {float(loss.detach()):.4f}", flush=True)
        bad += 0 if ok else 1
    except Exception as e:                                        # report every failure, keep going
        bad += 1
        print(f"{name:34s} FAILED: {type(e).__name__}: {str(e)[:160]}", flush=True)
print("defaults_sweep", "OK" if bad == 0 else f"{bad} FAILURES")

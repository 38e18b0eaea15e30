"""Host-side (launch) time of one bench step vs its device time: tells whether the step is host-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
import bench
from src.models.R2Plus1D import R2Plus1DClassifier
from src.loss import FocalLoss
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = R2Plus1DClassifier((3, bench.T, bench.S, bench.S), 2, bench.LAYERS, False, bench.ALPHA).to(dev).train()
loss_fn = FocalLoss(torch.ones(2), 2.0)
opt = torch.optim.AdamW(model.parameters(), lr=2e-4, fused=True)
x, y = bench.synth_batch(dev, 1)
def seg(name, f, acc):
    t0 = time.perf_counter(); r = f(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
for it in range(8):
    acc = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    seg('zero_grad', lambda: opt.zero_grad(set_to_none=True), acc)
    logits = seg('forward', lambda: model(x), acc)
    loss = seg('loss', lambda: loss_fn(logits, y), acc)
    seg('backward', lambda: loss.backward(), acc)
    seg('clip', lambda: torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0), acc)
    seg('adamw', lambda: opt.step(), acc)
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); total = time.perf_counter() - t0
    if it >= 3:
        print('host %.2f ms, host+device %.2f ms | ' % (host * 1e3, total * 1e3) + ' '.join('%s %.2f' % (k, v * 1e3) for k, v in acc.items()))

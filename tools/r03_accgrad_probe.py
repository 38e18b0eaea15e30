"""Which AccumulateGrad nodes survive a GraphedBranch capture, and who holds them?     python tools/r03_accgrad_probe.py"""
import gc, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
import src.models.fusion as fu
from src.models.fusion import FusionGB
from src.models.R2Plus1D import R2Plus1DClassifier
from src.models.transformer import Transformer

torch.manual_seed(21)
vis = R2Plus1DClassifier(input_size=(3, 5, 24, 24), num_classes=2, layer_sizes=[1, 1, 1, 1], alpha=0.01)
ts = Transformer(n_features=6, kernel_size=3, feature_dims=16, max_len=5, n_layers=1, n_heads=2, dim_feedforward=24, dropout=0.0, cls_dims=12, n_classes=2)
m = FusionGB(2, vis, ts).cuda().train()
for mod in m.modules():
    if type(mod).__name__ == "NoiseLayer":
        mod.std = 0.0
fu._GRAPH_BRANCH = True
torch.set_warn_always(True)
names = {id(p): k for k, p in m.named_parameters()}


def acc_ids():
    """id of the live AccumulateGrad node of every parameter (asking for it creates one where none is alive: report separately)"""
    out = {}
    for o in gc.get_objects():
        try:
            if type(o).__name__ == "AccumulateGrad":
                out[names.get(id(o.variable), "?")] = id(o)
        except Exception:
            pass
    return out


import traceback
_seen = set()
def _show(message, category, filename, lineno, file=None, line=None):
    if "AccumulateGrad node's stream" in str(message):
        st = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack() if "site-packages" not in f.filename and "dist-packages" not in f.filename or "graphs.py" in f.filename]
        key = tuple(st[-6:])
        if key not in _seen:
            _seen.add(key); print("   mismatch warning raised under:", " <- ".join(reversed(st[-7:])))
warnings.showwarning = _show

for step in range(3):
    xv, xt = torch.randn(4, 3, 5, 24, 24, device="cuda"), torch.randn(4, 5, 6, device="cuda")
    for p in m.parameters():
        p.grad = None
    caught = []
    if True:
        warnings.simplefilter("always")
        outs = m(xv, xt)
        alive_fwd = acc_ids()
        loss = sum(o.square().sum() for o in outs)
        loss.backward()
        torch.cuda.synchronize()
    nw = sum("AccumulateGrad node's stream" in str(w.message) for w in caught)
    del outs, loss
    gc.collect()
    alive_after = acc_ids()
    gbr = m.__dict__.get("_md_ts_graph")
    print(f"step {step}: mismatch warnings {nw}; released {getattr(gbr, 'released', None)}; AccumulateGrad nodes alive after the step: {len(alive_after)}",
          sorted(alive_after)[:6])
    if alive_after:
        k = sorted(alive_after)[0]
        node = [o for o in gc.get_objects() if type(o).__name__ == "AccumulateGrad" and id(o) == alive_after[k]][0]
        refs = gc.get_referrers(node)
        print("   referrers of", k, ":", [type(r).__name__ for r in refs][:8])
        del node, refs

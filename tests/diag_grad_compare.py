"""Per-parameter gradient deviation of the HIP path vs the oracle in fp32 and fp64 for one golden fixture:\n    python tests/diag_grad_compare.py r2p1d_1111_s2"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import numpy as np, torch
from oracle import losses as ol, r2plus1d as orc, step as ostep
from src.models.R2Plus1D import R2Plus1DClassifier
from src.loss import FocalLoss
tag = sys.argv[1] if len(sys.argv)>1 else 'r2p1d_tiny_a001'
g = np.load(os.path.join(ROOT, 'tests', 'golden', tag + '.npz'))
ls=[int(v) for v in g['layer_sizes']]; B,T,S,alpha,seed=int(g['B']),int(g['T']),int(g['S']),float(g['alpha']),int(g['seed'])
params,bufs=orc.synth_state(ls,seed,alpha)
x=orc.synth_clip(B,T,S,seed); y=orc.synth_labels(B,seed); w=torch.from_numpy(g['weight']); gamma=float(g['gamma'])
# fp32 oracle and fp64 oracle
_,_,g32=ostep.r2plus1d_loss_and_grads(x,y,params,bufs,ls,alpha,lambda o,t: ol.focal_loss(o,t,w,gamma))
p64={k:v.double() for k,v in params.items()}; b64={k:(v.double() if v.dtype==torch.float32 else v.clone()) for k,v in orc.synth_state(ls,seed,alpha)[1].items()}
_,_,g64=ostep.r2plus1d_loss_and_grads(x.double(),y,p64,b64,ls,alpha,lambda o,t: ol.focal_loss(o,t,w.double(),gamma))
model=R2Plus1DClassifier((3,T,S,S),2,ls,False,alpha)
sd=dict(params); sd.update(orc.synth_state(ls,seed,alpha)[1]); model.load_state_dict(sd)
model.cuda().train()
loss=FocalLoss(w,gamma)(model(x.cuda()),y.cuda()); loss.backward(); torch.cuda.synchronize()
named=dict(model.named_parameters())
print('%-60s %10s %10s %10s'%('param','hip-vs-64','cpu32-vs-64','hip-vs-32'))
for k in named:
    a=named[k].grad.cpu().double(); r=g64[k]; c=g32[k].double()
    sc=float(r.abs().max())+1e-30
    print('%-60s %10.2e %10.2e %10.2e'%(k, float((a-r).abs().max())/sc, float((c-r).abs().max())/sc, float((a-c).abs().max())/sc))

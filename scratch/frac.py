import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import losses as ol, r2plus1d as orc, step as ostep
ls=[1,2,2,1]; B,T,S,alpha,seed=5,6,48,1.0,2
w=torch.tensor([0.6,1.4])
def run(dt):
    params,bufs=orc.synth_state(ls,seed,alpha)
    params={k:v.to(dt) for k,v in params.items()}; bufs={k:(v.to(dt) if v.is_floating_point() else v) for k,v in bufs.items()}
    x=orc.synth_clip(B,T,S,seed).to(dt); y=orc.synth_labels(B,seed)
    return ostep.r2plus1d_loss_and_grads(x,y,params,bufs,ls,alpha,lambda o,t: ol.focal_loss(o,t,w.to(dt),2.0))[2]
g32=run(torch.float32); g64=run(torch.float64)
for k in g32:
    a=g32[k].double(); b=g64[k]; sc=float(b.abs().max())
    e=(a-b).abs()/sc
    if e.max()>2e-4: print('%-58s max %.1e  frac>1e-3: %.4f  frac>3e-4: %.4f  relL2 %.1e'%(k[11:], e.max(), float((e>1e-3).float().mean()), float((e>3e-4).float().mean()), float((a-b).norm()/b.norm())))

import sys; sys.path.insert(0,'.')
import numpy as np, torch, torch.nn.functional as F
from oracle import r2plus1d as orc
ls=[1,2,2,1]; B,T,S,alpha,seed=5,6,48,1.0,2
params,bufs=orc.synth_state(ls,seed,alpha)
x=orc.synth_clip(B,T,S,seed)
# instrument run_unit to print stats of raw outputs
orig=orc.run_unit
def ru(xx,u,sd,bufs,training):
    y=F.conv3d(xx, sd[u.name+'.conv.weight'], None, u.stride, u.padding)
    v=y.var(dim=(0,2,3,4),unbiased=False); m=y.mean(dim=(0,2,3,4))
    n=y.numel()//y.shape[1]
    print('%-50s n=%6d var[min %.3e med %.3e] |mean|/std max %.2f'%(u.name[11:], n, v.min(), v.median(), (m.abs()/v.sqrt()).max()))
    return orig(xx,u,sd,bufs,training)
orc.run_unit=ru
orc.classifier_forward(x,params,bufs,ls,alpha,True)

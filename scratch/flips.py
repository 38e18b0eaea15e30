import sys; sys.path.insert(0,'.')
import numpy as np, torch, torch.nn.functional as F
from oracle import r2plus1d as orc
ls=[1,2,2,1]; B,T,S,alpha,seed=5,6,48,1.0,2
def run(dt):
    params,bufs=orc.synth_state(ls,seed,alpha)
    params={k:v.to(dt) for k,v in params.items()}; bufs={k:(v.to(dt) if v.is_floating_point() else v) for k,v in bufs.items()}
    x=orc.synth_clip(B,T,S,seed).to(dt)
    pres={}
    def ru(xx,u,sd,bufs,training):
        y=F.conv3d(xx, sd[u.name+'.conv.weight'], None, u.stride, u.padding)
        y=orc._bn(y,sd,bufs,u.name+'.bn',training)
        pres[u.name]=y.detach().double()
        return F.leaky_relu(y,u.slope)
    orc.run_unit=ru
    orc.classifier_forward(x,params,bufs,ls,alpha,True)
    return pres
p32=run(torch.float32); p64=run(torch.float64)
for k in p32:
    a,b=p32[k],p64[k]
    flips=int(((a>0)!=(b>0)).sum()); 
    print('%-50s n=%8d flips=%4d maxabs diff=%.2e  min|pre64|=%.2e'%(k[11:],a.numel(),flips,float((a-b).abs().max()),float(b.abs().min())))

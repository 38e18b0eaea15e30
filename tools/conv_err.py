"""Relative error of the conv forward (exact-fp32 mode vs split mode) against an fp64 convolution."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch, numpy as np, torch.nn.functional as F
from src import ops
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_ops_gpu import cl, uncl
torch.manual_seed(0)
for name,Cin,Cout,k,s,p,shape,scale in [("sp3x3",32,72,(1,3,3),(1,1,1),(0,1,1),(2,3,32,32),1.0),("tmp3",72,32,(3,1,1),(1,1,1),(1,0,0),(2,5,16,16),1.0),("small_act",64,144,(1,3,3),(1,1,1),(0,1,1),(2,3,16,16),0.01),("big_act",64,144,(1,3,3),(1,1,1),(0,1,1),(2,3,16,16),200.0),("stem",3,45,(1,7,7),(1,2,2),(0,3,3),(2,2,64,64),100.0)]:
    N,T,H,W=shape
    x=torch.randn(N,Cin,T,H,W)*scale; w=torch.randn(Cout,Cin,*k)/np.sqrt(Cin*k[0]*k[1]*k[2])
    y64=F.conv3d(x.double(),w.double(),None,s,p)
    y32=F.conv3d(x,w,None,s,p).double()
    d=ops.make_desc(N,T,H,W,Cin,Cout,k,s,p)
    for exact in (True,False):
        ops.set_exact_fp32(exact)
        wf,_=ops.pack_weights(d,w.cuda(),False)
        yg,_=ops.conv_fwd(d,ops.view(cl(x).cuda()),wf,'cuda:0',False)
        e=float((uncl(yg.cpu(),Cout).double()-y64).abs().max()/y64.abs().max())
        print(name, 'exact' if exact else 'split', 'max rel err vs fp64 %.2e'%e, ' (cpu fp32: %.2e)'%float((y32-y64).abs().max()/y64.abs().max()))
    ops.set_exact_fp32(False)

"""One op of one BASELINE layer, a few launches, for rocprofv3 --pmc runs:  python3 tools/pmc_layer.py c1s fwd|dgrad|wgrad"""
import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch
from src import ops
L = {
 'c1s': (32,72,(1,3,3),(1,1,1),(0,1,1),(8,21,64,64)),
 'c1t': (72,32,(3,1,1),(1,1,1),(1,0,0),(8,21,64,64)),
 'c3s': (64,144,(1,3,3),(1,1,1),(0,1,1),(8,11,32,32)),
 'c3t': (144,64,(3,1,1),(1,1,1),(1,0,0),(8,11,32,32)),
 'stem': (3,45,(1,7,7),(1,2,2),(0,3,3),(8,21,128,128)),
 'c3d': (32,115,(1,3,3),(1,2,2),(0,1,1),(8,21,64,64)),
}
name, op = sys.argv[1], sys.argv[2]
Cin,Cout,k,s,p,(N,T,H,W) = L[name]
d = ops.make_desc(N,T,H,W,Cin,Cout,k,s,p)
x = torch.randn(N,T,H,W,ops.cpad(Cin),device='cuda'); w = torch.randn(Cout,Cin,*k,device='cuda')*0.05
sc = torch.rand(ops.cpad(Cin),device='cuda')+0.5; sh = torch.randn(ops.cpad(Cin),device='cuda')*0.1
wf,wd = ops.pack_weights(d,w)
dy = torch.randn(N,d.To,d.Ho,d.Wo,ops.cpad(Cout),device='cuda')
v = ops.view(x,sc,sh,0.01)
f = {'fwd': lambda: ops.conv_fwd(d,v,wf,'cuda:0',True), 'dgrad': lambda: ops.conv_dgrad(d,dy,wd), 'wgrad': lambda: ops.conv_wgrad(d,v,dy)}[op]
for _ in range(4): f()
torch.cuda.synchronize()

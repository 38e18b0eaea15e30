"""Per-layer timing of the conv kernels at the BASELINE shapes (run on the GPU box):  python tools/layer_bench.py [c1s c1t c3s]
MD_DBG=<bits> ablates phases of the patch kernels (see conv_patch.hip); timings below ~50 us are host-launch bound."""
import sys, os, time; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch, numpy as np
from src import ops
L = {
 'c1s': (32,72,(1,3,3),(1,1,1),(0,1,1),(8,21,64,64)),
 'c1t': (72,32,(3,1,1),(1,1,1),(1,0,0),(8,21,64,64)),
 'c3s': (64,144,(1,3,3),(1,1,1),(0,1,1),(8,11,32,32)),
 'stem': (3,45,(1,7,7),(1,2,2),(0,3,3),(8,21,128,128)),
 'c3d': (32,115,(1,3,3),(1,2,2),(0,1,1),(8,21,64,64)),
}
which = sys.argv[1:] or list(L)
for name in which:
    Cin,Cout,k,s,p,(N,T,H,W) = L[name]
    d = ops.make_desc(N,T,H,W,Cin,Cout,k,s,p)
    x = torch.randn(N,T,H,W,ops.cpad(Cin),device='cuda'); w = torch.randn(Cout,Cin,*k,device='cuda')*0.05
    sc = torch.rand(ops.cpad(Cin),device='cuda')+0.5; sh = torch.randn(ops.cpad(Cin),device='cuda')*0.1
    wf,wd = ops.pack_weights(d,w)
    dy = torch.randn(N,d.To,d.Ho,d.Wo,ops.cpad(Cout),device='cuda')
    v = ops.view(x,sc,sh,0.01)
    def timeit(f, n=20):
        for _ in range(3): f()
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
    fl = 2.0*N*d.To*d.Ho*d.Wo*Cout*Cin*k[0]*k[1]*k[2]
    by = 4.0*(x.numel()+dy.numel())
    tf = timeit(lambda: ops.conv_fwd(d,v,wf,'cuda:0',True))
    td = timeit(lambda: ops.conv_dgrad(d,dy,wd))
    tw = timeit(lambda: ops.conv_wgrad(d,v,dy))
    print(f"{name} dbg={os.environ.get('MD_DBG','0')}: fwd {tf:7.1f} us ({fl/tf/1e6:6.1f} TF, {by/tf/1e3:6.0f} GB/s)  dgrad {td:7.1f} us  wgrad {tw:7.1f} us  | io floor {by/5e6:5.1f} us")

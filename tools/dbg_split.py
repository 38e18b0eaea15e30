import sys, os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'disruption-prediciton-based-on-multimodal-deep-learning_amd'))
import torch, numpy as np, struct
from src import ops
torch.manual_seed(0)
C=32; rows=4096
for scale_mode in (0, 1):
    y=torch.randn(rows,C,device='cuda'); dA=torch.randn(rows,C,device='cuda')
    if scale_mode: dA = dA * torch.logspace(-6, 2, C, device='cuda').view(1,-1)
    st=torch.zeros(4,C,device='cuda'); st[0]=0.1; st[1]=1.3; st[2]=torch.rand(C,device='cuda')+0.5; st[3]=torch.randn(C,device='cuda')*0.3
    coef=torch.randn(2,C,device='cuda')*0.01
    yv=ops.view(y.view(1,1,1,rows,C),st[2],st[3],0.01)
    d32=ops.bn_apply_fmt(dA.view(1,1,1,rows,C).contiguous(),yv,st,coef,C,False).view(-1,8)
    dsp=ops.bn_apply_fmt(dA.view(1,1,1,rows,C).contiguous(),yv,st,coef,C,True)
    hl=dsp.view(torch.bfloat16).view(-1,2,8).float()
    hi_ref=d32.bfloat16().float()
    bad=(hl[:,0]!=hi_ref)
    print('mode', scale_mode, 'hi mismatches', int(bad.sum()), 'of', bad.numel())
    rec=hl[:,0]+hl[:,1]
    print('  max |hi+lo - d32| / |d32|', float(((rec-d32).abs()/d32.abs().clamp_min(1e-30)).max()))
    for i,j in bad.nonzero()[:6].tolist():
        v=d32[i,j].item()
        print('  ', v, hex(struct.unpack('<I',struct.pack('<f',v))[0]), 'kernel hi', hl[i,0,j].item(), 'torch hi', hi_ref[i,j].item(), 'lo', hl[i,1,j].item())

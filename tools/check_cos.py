import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd.nn import _ops
torch.manual_seed(0)
x = torch.cat([torch.rand(200000)*10, torch.rand(200000)*3e6, torch.rand(200000)*2.1e9, -torch.rand(100000)*1e7, torch.tensor([0.,1e-30,3.14159265,1.5707963,2147483648.])]).cuda()
w = torch.ones(1, device='cuda'); b = torch.zeros(1, device='cuda')
got = _ops.time2vec(x, w, b)[:,0].cpu().double()
ref = torch.cos(x.cpu().double())
err = (got-ref).abs()
print('max abs err vs f64 cos:', err.max().item(), 'at x=', x.cpu()[err.argmax()].item(), ' mean', err.mean().item())

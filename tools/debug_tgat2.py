import os, sys
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import _native
from oracle import tgat_ref
DEV='cuda'
torch.manual_seed(0)
lib=_native.load()
R,k,d,D,T,H=64,1,1,172,100,2
C=d+D+T
st=torch.randint(1_000_000,2_600_000,(R,),device=DEV)
nt=(st[:,None]-torch.randint(1,900_000,(R,k),device=DEV)).clamp(min=0)
nid=torch.randint(0,1000,(R,k),dtype=torch.int32,device=DEV)
ex=torch.rand(R,k,D,device=DEV); nbrf=torch.randn(R,k,d,device=DEV)
qf=torch.randn(R,H,C,device=DEV)*0.1
for jit in (0.0, 0.03):
    w=(torch.from_numpy((1/10**np.linspace(0,9,T))).float()+jit*torch.randn(T)).to(DEV); b=(jit*torch.randn(T)).to(DEV)
    zbar=torch.empty(R,H,C,device=DEV)
    _native.check(lib.tgmx_tgat_attn_reduce(qf.data_ptr(), nbrf.data_ptr(), d, ex.data_ptr(), D, st.data_ptr(), nt.data_ptr(), nid.data_ptr(), w.data_ptr(), b.data_ptr(), 0,0, T,H,k,R, 0.1, zbar.data_ptr(), _native.stream_ptr()),'x')
    tf=tgat_ref.time2vec((st[:,None]-nt).cpu(), w.cpu().view(T,1), b.cpu())
    Z=torch.cat([nbrf.cpu(), ex.cpu(), tf],-1)[:,0]   # [R,C]
    z=zbar.cpu()
    for h in range(H):
        e=(z[:,h]-Z).abs()
        print('jit',jit,'h',h,'nbr',e[:,:d].max().item(),'edge',e[:,d:d+D].max().item(),'time',e[:,d+D:].max().item())
    # time2vec kernel
    from tgm_amd.nn import _ops
    t2=_ops.time2vec((st[:,None]-nt), w, b).cpu()
    print('   time2vec kernel vs cpu', (t2-tf).abs().max().item())
    dtf=(st[:,None]-nt).float()
    arg_gpu = torch.addcmul(b, dtf[...,None], w)  # not fma necessarily
    print('   cos(cpu-arg) on gpu', (torch.cos(torch.nn.functional.linear(dtf.cpu()[...,None], w.cpu().view(T,1), b.cpu()).to(DEV)).cpu()-tf).abs().max().item())

import os, sys
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import _native
from tgm_amd.nn import _ops, TGAT
from oracle import tgat_ref
DEV='cuda'
torch.manual_seed(0)
lib=_native.load()
def rel(a,b): 
    a=a.cpu(); b=b.cpu(); return ((a-b).abs()/(1e-5*b.abs().clamp(min=1))).max().item()
for R in (500, 5000, 12600, 40000):
    k,d,D,T,H=20,1,172,100,2
    O=d+T+1; dh=O//H; C=d+D+T
    st=torch.randint(1_000_000,2_600_000,(R,),device=DEV)
    nt=(st[:,None]-torch.randint(1,900_000,(R,k),device=DEV)).clamp(min=0)
    nid=torch.randint(0,1000,(R,k),dtype=torch.int32,device=DEV); nid[torch.rand(R,k,device=DEV)<0.5]=-1; nt[nid==-1]=0
    ex=torch.rand(R,k,D,device=DEV); ex[nid==-1]=0
    nbrf=torch.randn(R,k,d,device=DEV)
    qf=torch.randn(R,H,C,device=DEV)*0.1
    w=(torch.from_numpy((1/10**np.linspace(0,9,T))).float()+0.03*torch.randn(T)).to(DEV); b=(0.03*torch.randn(T)).to(DEV)
    zbar=torch.empty(R,H,C,device=DEV)
    _native.check(lib.tgmx_tgat_attn_reduce(qf.data_ptr(), nbrf.data_ptr(), d, ex.data_ptr(), D, st.data_ptr(), nt.data_ptr(), nid.data_ptr(), w.data_ptr(), b.data_ptr(), 0,0, T,H,k,R, float(dh)**-0.5, zbar.data_ptr(), _native.stream_ptr()),'x')
    # CPU ref
    tf=tgat_ref.time2vec((st[:,None]-nt).cpu(), w.cpu().view(T,1), b.cpu())
    Z=torch.cat([nbrf.cpu(), ex.cpu(), tf],-1)
    A=torch.einsum('bhc,bkc->bhk', qf.cpu(), Z)*dh**-0.5
    A=A.masked_fill(~(nid.cpu()!=-1)[:,None,:], -1e10).softmax(-1)
    zref=torch.einsum('bhk,bkc->bhc',A,Z)
    e=(zbar.cpu()-zref).abs()
    print('R',R,'attn_reduce worst', rel(zbar,zref), 'max abs', e.max().item(), 'bad rows', (e.amax((1,2))>1e-4).sum().item(), 'first bad', (e.amax((1,2))>1e-4).nonzero()[:5].flatten().tolist())
    # gemm check
    A_=torch.randn(R,102,device=DEV); B_=torch.randn(102,102,device=DEV); out=torch.empty(R,102,device=DEV)
    _ops.sgemm_nt(A_,B_,out); print('   gemm worst', rel(out, A_.cpu()@B_.cpu().T))

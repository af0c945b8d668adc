import os, sys
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import _native
DEV='cuda'
torch.manual_seed(0)
libpath=os.environ.get('TGMX_LIB')
if libpath:
    _native._lib=None; lib=_native.load(libpath)
else:
    lib=_native.load()
R,k,d,D,T,H=12000,20,1,172,100,2
C=d+D+T
st=torch.randint(1_000_000,2_600_000,(R,),device=DEV)
nt=(st[:,None]-torch.randint(1,900_000,(R,k),device=DEV)).clamp(min=0)
nt[torch.rand(R,device=DEV)<0.75]=0  # three quarters of the layer-1 rows are padded seeds: one time delta for all slots
nid=torch.randint(0,1000,(R,k),dtype=torch.int32,device=DEV)
ex=torch.rand(R,k,D,device=DEV); nbrf=torch.randn(R,k,d,device=DEV)
qf=torch.randn(R,H,C,device=DEV)*0.1
w=torch.from_numpy((1/10**np.linspace(0,9,T))).float().to(DEV); b=torch.zeros(T,device=DEV)
tfeat=torch.rand(R,k,T,device=DEV)
zbar=torch.empty(R,H,C,device=DEV)
def run(tf):
    return lib.tgmx_tgat_attn_reduce(qf.data_ptr(), nbrf.data_ptr(), d, ex.data_ptr(), D, st.data_ptr(), nt.data_ptr(), nid.data_ptr(), w.data_ptr(), b.data_ptr(), tf, 0, T,H,k,R, 0.1, 0, zbar.data_ptr(), 0, _native.stream_ptr())
for name, tf in (('cos in kernel', 0),):
    for _ in range(3): run(tf)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(tf)
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1)/20*1000, 'us')

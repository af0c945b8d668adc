import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd.nn import _ops
DEV='cuda'
def bench(M,N,K,lda=None,batch=1):
    lda = lda or K
    A=torch.randn(M,lda,device=DEV)[:, :K] if lda!=K else torch.randn(M,K,device=DEV)
    B=torch.randn(N,K,device=DEV); C=torch.empty(M,N,device=DEV)
    for _ in range(3): _ops.sgemm_nt(A,B,C)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    n=20; e0.record()
    for _ in range(n): _ops.sgemm_nt(A,B,C)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/n*1000
    ref=(A.double()@B.double().T)
    err=((C.double()-ref).abs().max()/ref.abs().max()).item()
    print(f'M={M:6d} N={N:4d} K={K:4d} lda={lda:4d}: {us:8.1f} us  {2*M*N*K/us/1e6:7.2f} TF/s  relerr {err:.1e}')
bench(4096,4096,4096)
bench(12600,172,172)
bench(12600,172,103)
bench(12600,102,102)
bench(12600,104,104)
bench(12600,273,51)
bench(600,544,444)
bench(600,172,516)
bench(65536,256,256)

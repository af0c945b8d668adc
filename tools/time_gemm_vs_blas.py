#!/usr/bin/env python
"""tgmx_sgemm_nt against the vendor library (torch.mm -> rocBLAS / hipBLASLt, fp32) on the many-row shapes of the training step and of cfg 3:
what a tuned plain GEMM gets on these shapes, i.e. how far the hand-written operands-from-global kernel is from a library kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd.nn import _ops
DEV = 'cuda'
torch.backends.cuda.matmul.allow_tf32 = False


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


def bench(M, N, K, label=''):
    A = torch.randn(M, K, device=DEV); B = torch.randn(N, K, device=DEV); C = torch.empty(M, N, device=DEV)
    ours = t(lambda: _ops.sgemm_nt(A, B, C))
    Bt = B.t()
    lib = t(lambda: torch.mm(A, Bt, out=C))
    ref = A.double() @ B.double().T
    _ops.sgemm_nt(A, B, C); e1 = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    torch.mm(A, Bt, out=C); e2 = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    gf = 2 * M * N * K / 1e6
    print(f'{label:16s} M={M:6d} N={N:4d} K={K:4d}: ours {ours:7.1f} us {gf / ours:6.1f} TF/s err {e1:.1e} | library {lib:7.1f} us {gf / lib:6.1f} TF/s err {e2:.1e}', flush=True)


for M in (12600,):
    bench(M, 172, 172, 'fc2'); bench(M, 172, 276, 'fc1'); bench(M, 104, 104, 'W_O'); bench(M, 276, 52, 'qf/head'); bench(M, 448, 272, 'L2-ish')
bench(8000, 300, 316, 'cfg3 gi'); bench(8000, 300, 100, 'cfg3 gh'); bench(8000, 400, 100, 'cfg3 proj'); bench(13000, 100, 116, 'cfg3 eproj')
bench(600, 172, 172, 'small'); bench(4096, 4096, 4096, 'square')

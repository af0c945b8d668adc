#!/usr/bin/env python
"""Time tgmx_sgemm_nt on the shapes of the TGAT forward (padded operand layouts) and a large square case."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd.nn import _ops
DEV = 'cuda'


def bench(M, N, K, label=''):
    pad = lambda x: (x + 3) // 4 * 4
    A = torch.randn(M, pad(K), device=DEV)[:, :K]
    B = torch.randn(N, pad(K), device=DEV)[:, :K]
    C = torch.empty(M, pad(N), device=DEV)[:, :N]
    for _ in range(3):
        _ops.sgemm_nt(A, B, C)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        _ops.sgemm_nt(A, B, C)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1000
    ref = A.double() @ B.double().T
    err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    print(f'{label:14s} M={M:6d} N={N:4d} K={K:4d}: {us:8.1f} us  {2 * M * N * K / us / 1e6:7.2f} TF/s  relerr {err:.1e}', flush=True)


print('KS', os.environ.get('TGMX_GEMM_KS'), 'SPLIT', os.environ.get('TGMX_GEMM_SPLIT'))
bench(12600, 102, 102, 'L1 W_O')
bench(12600, 273, 51, 'L1 qf/head')
bench(12600, 51, 273, 'L1 V/head')
bench(12600, 172, 103, 'L1 fc1')
bench(12600, 172, 172, 'L1 fc2')
bench(600, 272, 272, 'L2 W_O')
bench(600, 444, 136, 'L2 qf/head')
bench(600, 136, 444, 'L2 V/head')
bench(600, 172, 444, 'L2 fc1')
bench(600, 172, 172, 'L2 fc2')
bench(4096, 4096, 4096, 'square')

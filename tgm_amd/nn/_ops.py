"""Thin typed wrappers over the TGAT entry points of libtgm_amd.so (forward only)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from .. import _native


def _f32c(t: Tensor, what: str) -> Tensor:
    _native.require_device(t, what)
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def time2vec(t: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """cos(fma(float32(t), w, b)) for int64 / float timestamps of any shape -> [..., T]."""
    _native.require_device(t, 'time2vec input')
    T = w.numel()
    if t.dtype in (torch.int64,):
        x, is64 = t.contiguous(), 1
    else:
        x, is64 = _f32c(t, 'time2vec input'), 0
    out = torch.empty(x.shape + (T,), dtype=torch.float32, device=x.device)
    lib = _native.load()
    _native.check(lib.tgmx_time2vec(x.data_ptr(), is64, _f32c(w, 'w').data_ptr(), _f32c(b, 'b').data_ptr(), T, x.numel(), out.data_ptr(), _native.stream_ptr()), 'tgmx_time2vec')
    return out


def gather_rows(table: Tensor, idx: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """table[idx] with Python-style negative wrap, idx int32 [n] -> [n, dim]."""
    table = _f32c(table, 'feature table')
    n, dim = idx.numel(), table.shape[1]
    if out is None:
        out = torch.empty((n, dim), dtype=torch.float32, device=table.device)
    lib = _native.load()
    _native.check(lib.tgmx_gather_rows(table.data_ptr(), table.shape[0], dim, idx.data_ptr(), n, out.data_ptr(), out.stride(0), _native.stream_ptr()), 'tgmx_gather_rows')
    return out


def sgemm_nt(A: Tensor, B: Tensor, out: Tensor, bias: Optional[Tensor] = None, relu: bool = False, M: Optional[int] = None,
             N: Optional[int] = None, K: Optional[int] = None, batch: int = 1, sA: int = 0, sB: int = 0, sC: int = 0) -> Tensor:  # fmt: skip
    """out[M, N] = act(A[M, K] @ B[N, K].T + bias); 2-D (possibly column-sliced) row-major views."""
    M = A.shape[0] if M is None else M
    N = B.shape[0] if N is None else N
    K = A.shape[1] if K is None else K
    lib = _native.load()
    _native.check(
        lib.tgmx_sgemm_nt(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), out.data_ptr(), out.stride(0), M, N, K,
                          _native.ptr(bias), 1 if relu else 0, batch, sA, sB, sC, _native.stream_ptr()),
        'tgmx_sgemm_nt',
    )  # fmt: skip
    return out

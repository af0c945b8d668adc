"""``TGCN`` -- temporal graph convolutional GRU cell (tgm/nn/encoder/tgcn.py:8-157) on HIP kernels.

Parameter names match the reference / PyG (``conv_{u,r,c}.lin.weight``, ``conv_{u,r,c}.bias``,
``linear_{u,r,c}.{weight,bias}``).  ``GCNConv`` is third-party to the reference (torch_geometric);
it is implemented from its published definition (parity unpinned upstream).  The three graph
convolutions share one dense normalised adjacency and run as two exact-fp32 MFMA GEMMs
(``A_hat @ (X [W_u|W_r|W_c])``); snapshot graphs on this path are small (tgbn-trade: 255 nodes).
With gradients enabled the same arithmetic runs inside ``torch.autograd.Function``s with a hand-written backward
(``nn/_tgcn_train.py``: the reference trains this cell, examples/nodeproppred/tgcn.py:92).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import _native
from . import _ops
from ._paramver import TransientCaches

_MAX_DENSE_NODES = 16384  # A_hat is dense: 16384^2 floats = 1 GiB


class GCNConv(nn.Module):
    """x' = D^-1/2 (A + I) D^-1/2 x W^T + b  (PyG parameter layout: ``lin.weight`` [out, in], ``bias`` [out])."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False, cached: bool = False, add_self_loops: bool = True) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached, self.add_self_loops = improved, cached, add_self_loops
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        bound = math.sqrt(6.0 / (in_channels + out_channels))  # glorot
        nn.init.uniform_(self.lin.weight, -bound, bound)

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None) -> Tensor:
        x = _ops._f32c(x, 'x')
        A = normalized_adjacency(edge_index, edge_weight, x.shape[0], 2.0 if self.improved else 1.0, self.add_self_loops)
        if torch.is_grad_enabled() and (x.requires_grad or self.lin.weight.requires_grad or self.bias.requires_grad):
            from ._tgcn_train import AdjMatmulFn
            from ._tgn_train import LinearFn

            return AdjMatmulFn.apply(A, LinearFn.apply(x, self.lin.weight, None)) + self.bias
        xwt = torch.empty((self.out_channels, x.shape[0]), dtype=torch.float32, device=x.device)
        _ops.sgemm_nt(self.lin.weight.detach(), x, xwt)  # (X W^T)^T = W X^T
        out = torch.empty((x.shape[0], self.out_channels), dtype=torch.float32, device=x.device)
        return _ops.sgemm_nt(A, xwt, out, bias=self.bias.detach(), K=x.shape[0])


def normalized_adjacency(edge_index: Tensor, edge_weight: Optional[Tensor], N: int, fill: float, add_self_loops: bool) -> Tensor:
    """Dense D^-1/2 (A + I) D^-1/2, rows padded to a multiple of 4 floats (16-byte aligned GEMM operand)."""
    if N > _MAX_DENSE_NODES:
        raise NotImplementedError(f'tgm_amd GCNConv builds a dense adjacency; {N} nodes exceed {_MAX_DENSE_NODES}')
    _native.require_device(edge_index, 'edge_index')
    lib = _native.load()
    dev = edge_index.device
    ei = edge_index.to(torch.int64)
    src, dst = ei[0].contiguous(), ei[1].contiguous()
    w = None if edge_weight is None else _ops._f32c(edge_weight, 'edge_weight')
    ld = (N + 3) // 4 * 4
    A = torch.empty((N, ld), dtype=torch.float32, device=dev)
    ws = torch.empty(2 * N, dtype=torch.float32, device=dev)
    _native.check(
        lib.tgmx_gcn_norm_dense(src.data_ptr(), dst.data_ptr(), _native.ptr(w), src.numel(), N, float(fill), 1 if add_self_loops else 0,
                                A.data_ptr(), ld, ws.data_ptr(), _native.stream_ptr()),
        'tgmx_gcn_norm_dense',
    )  # fmt: skip
    return A[:, :N]


class TGCN(TransientCaches, nn.Module):
    def __init__(self, in_channels: int, out_channels: int, improved: bool = False, cached: bool = False, add_self_loops: bool = True) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.improved, self.cached, self.add_self_loops = improved, cached, add_self_loops
        mk = lambda: GCNConv(in_channels, out_channels, improved=improved, cached=cached, add_self_loops=add_self_loops)
        self.conv_c, self.linear_c = mk(), nn.Linear(2 * out_channels, out_channels)
        self.conv_r, self.linear_r = mk(), nn.Linear(2 * out_channels, out_channels)
        self.conv_u, self.linear_u = mk(), nn.Linear(2 * out_channels, out_channels)

    def forward(self, node_x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None, H: Optional[Tensor] = None) -> Tensor:
        lib = _native.load()
        x = _ops._f32c(node_x, 'node_x')
        N, C, dev = x.shape[0], self.out_channels, x.device
        f32 = dict(dtype=torch.float32, device=dev)
        H = torch.zeros((N, C), **f32) if H is None else _ops._f32c(H, 'H')
        stream = _native.stream_ptr()
        grad = torch.is_grad_enabled() and (x.requires_grad or H.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not grad and os.environ.get('TGMX_TGCN_PY') is None:
            return self._forward_native(x, edge_index, edge_weight, H)
        A = normalized_adjacency(edge_index, edge_weight, N, 2.0 if self.improved else 1.0, self.add_self_loops)
        if grad:
            from ._tgcn_train import TGCNCellFn

            gates = (self.conv_u, self.conv_r, self.conv_c), (self.linear_u, self.linear_r, self.linear_c)
            return TGCNCellFn.apply(x, A, H, *[c.lin.weight for c in gates[0]], *[c.bias for c in gates[0]], *[l.weight for l in gates[1]],
                                    *[l.bias for l in gates[1]])  # fmt: skip
        # (TGMX_TGCN_PY=1, A/B: the same launches composed from Python, one ctypes call each)
        # the three convolutions share A_hat: G = A_hat @ (X [W_u | W_r | W_c]^T) + [b_u | b_r | b_c]
        W3 = torch.cat([self.conv_u.lin.weight.detach(), self.conv_r.lin.weight.detach(), self.conv_c.lin.weight.detach()])
        b3 = torch.cat([self.conv_u.bias.detach(), self.conv_r.bias.detach(), self.conv_c.bias.detach()])
        xwt = torch.empty((3 * C, N), **f32)
        _ops.sgemm_nt(W3, x, xwt)
        G = torch.empty((N, 3 * C), **f32)
        _ops.sgemm_nt(A, xwt, G, bias=b3, K=N)
        cat = torch.empty((N, 2 * C), **f32)
        pre = [torch.empty((N, C), **f32) for _ in range(3)]  # u, r, c pre-activations
        for g, (lin, gate) in enumerate(((self.linear_u, None), (self.linear_r, None), (self.linear_c, pre[1]))):
            _native.check(lib.tgmx_tgcn_concat(G[:, g * C :].data_ptr(), 3 * C, H.data_ptr(), _native.ptr(gate), C, N, cat.data_ptr(), stream), 'tgmx_tgcn_concat')
            _ops.sgemm_nt(cat, lin.weight.detach(), pre[g], bias=lin.bias.detach())
        out = torch.empty((N, C), **f32)
        _native.check(lib.tgmx_tgcn_output(pre[0].data_ptr(), pre[2].data_ptr(), H.data_ptr(), N * C, out.data_ptr(), stream), 'tgmx_tgcn_output')
        return out

    def _forward_native(self, x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], H: Tensor) -> Tensor:
        """The inference forward as ONE native call (``tgmx_tgcn_forward``): the launches of the Python-composed sequence above in the same order
        (identical results), the stacked GCN weights cached against the parameters' versions, the scratch kept between snapshots -- a
        255-node snapshot is 13 launches and was host-bound at ~160 us when every launch was its own ctypes call between torch ops."""
        from ._paramver import param_key

        if x.shape[0] > _MAX_DENSE_NODES:
            raise NotImplementedError(f'tgm_amd GCNConv builds a dense adjacency; {x.shape[0]} nodes exceed {_MAX_DENSE_NODES}')
        _native.require_device(edge_index, 'edge_index')
        lib = _native.load()
        N, C, dev = x.shape[0], self.out_channels, x.device
        d = self.__dict__
        key = param_key(self)
        if d.get('_tgmx_w3_key') != key:
            convs, lins = (self.conv_u, self.conv_r, self.conv_c), (self.linear_u, self.linear_r, self.linear_c)
            d['_tgmx_w3'] = (torch.cat([c.lin.weight.detach() for c in convs]).contiguous(), torch.cat([c.bias.detach() for c in convs]).contiguous(),
                             [_ops._f32c(l.weight.detach(), 'weight') for l in lins], [_ops._f32c(l.bias.detach(), 'bias') for l in lins])
            d['_tgmx_w3_key'] = key
        W3, b3, lw, lb = d['_tgmx_w3']
        ws = d.get('_tgmx_ws')
        ld = (N + 3) // 4 * 4
        up = lambda n: (n + 63) // 64 * 64  # 256-byte granules: every region a 16-byte aligned GEMM operand
        sizes = [up(N * ld), up(2 * N), up(3 * C * N), up(3 * C * N), up(2 * C * N), up(C * N), up(C * N), up(C * N)]
        need = sum(sizes)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = d['_tgmx_ws'] = torch.empty(need, dtype=torch.float32, device=dev)
        ei = edge_index if edge_index.dtype in (torch.int64, torch.int32) else edge_index.to(torch.int64)
        src, dst = ei[0], ei[1]
        if not src.is_contiguous():
            src = src.contiguous()
        if not dst.is_contiguous():
            dst = dst.contiguous()
        w = None if edge_weight is None else _ops._f32c(edge_weight, 'edge_weight')
        out = torch.empty((N, C), dtype=torch.float32, device=dev)
        a = d.get('_tgmx_args')
        if a is None:
            a = d['_tgmx_args'] = _native.TgcnFwd()
        base = ws.data_ptr()
        a.x, a.N, a.in_ch, a.C = x.data_ptr(), N, x.shape[1], C
        a.src, a.dst, a.edge_w, a.E = src.data_ptr(), dst.data_ptr(), _native.ptr(w), src.numel()
        a.idx32 = 1 if ei.dtype == torch.int32 else 0
        a.fill, a.add_self_loops = (2.0 if self.improved else 1.0), (1 if self.add_self_loops else 0)
        a.W3, a.b3 = W3.data_ptr(), b3.data_ptr()
        for g in range(3):
            a.lin_w[g], a.lin_b[g] = lw[g].data_ptr(), lb[g].data_ptr()
        a.H = H.data_ptr()
        ptrs, off = [], 0
        for n_ in sizes:
            ptrs.append(base + 4 * off)
            off += n_
        a.A, a.ldA, a.norm_ws, a.xwt, a.G, a.cat = ptrs[0], ld, ptrs[1], ptrs[2], ptrs[3], ptrs[4]
        for g in range(3):
            a.pre[g] = ptrs[5 + g]
        a.out = out.data_ptr()
        _native.check(lib.tgmx_tgcn_forward(ctypes.byref(a), _native.stream_ptr()), 'tgmx_tgcn_forward')
        return out

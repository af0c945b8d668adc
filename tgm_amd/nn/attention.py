"""``TemporalAttention`` -- multi-head attention of one query over k sampled neighbors.

Same constructor, parameters (``W_Q``, ``W_KV``, ``W_O``, ``layer_norm``) and forward
signature as tgm/nn/modules/attention.py:5-128, so state_dicts interchange.  The forward
runs on the HIP kernels of ``csrc/tgat.hip`` with the W_KV projection folded onto the
query / output side (q-length 1; see ``oracle/tgat_fold.py`` for the algebra).
Used stand-alone the module is forward-only (no autograd graph is recorded; training goes
through ``TGAT``, whose hand-written backward covers this layer).  In ``.train()`` mode the two
dropout sites of the reference (attention.py:119 on the attention weights, :126 on the W_O
output) are applied with counter-based masks (``tgmx_dropout_t``): seeded from
``torch.initial_seed()``, a fresh stream per call.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from .. import _native
from . import _ops


class TemporalAttention(nn.Module):
    def __init__(self, n_heads: int, node_dim: int, edge_dim: int, time_dim: int, dropout: float = 0.1) -> None:
        super().__init__()
        if any(x <= 0 for x in (n_heads, node_dim, edge_dim, time_dim)):
            raise ValueError('n_heads,node_dim,edge_dim,time_dim,out_dim must be > 0')
        out_dim = node_dim + time_dim
        self.pad_dim = 0
        if out_dim % n_heads != 0:
            self.pad_dim = n_heads - out_dim % n_heads
            out_dim += self.pad_dim
        self.n_heads, self.head_dim, self.out_dim = n_heads, out_dim // n_heads, out_dim
        self.node_dim, self.edge_dim, self.time_dim = node_dim, edge_dim, time_dim
        key_dim = node_dim + edge_dim + time_dim
        self.W_Q = nn.Linear(out_dim, out_dim, bias=False)
        self.W_KV = nn.Linear(key_dim, out_dim * 2, bias=False)
        self.W_O = nn.Linear(out_dim, out_dim)
        self.dropout = nn.Dropout(dropout)
        self.layer_norm = nn.LayerNorm(out_dim)

    # ------------------------------------------------------------------
    def _dropout_site(self, site: int):
        """tgmx_dropout_t of this call's dropout site (0: attention weights, 1: W_O output); None when inactive."""
        p = self.dropout.p if self.training else 0.0
        if not p:
            return None
        return _native.dropout_desc(p, torch.initial_seed(), self._drop_calls * 64 + site)

    def attend(
        self,
        rres: Tensor,  # [R, O] residual == query input
        nbrf: Tensor,  # [R, k, d]
        ex: Tensor,  # [R, k, D]
        k: int,
        *,
        seed_t: Optional[Tensor] = None,
        nbr_t: Optional[Tensor] = None,
        nbr_id: Optional[Tensor] = None,
        tw: Optional[Tensor] = None,
        tb: Optional[Tensor] = None,
        nbr_time_feat: Optional[Tensor] = None,
        mask: Optional[Tensor] = None,
        z0: Optional[Tensor] = None,
    ) -> Tensor:
        """LayerNorm(W_O attn + b_O + rres), optionally with ``z0`` appended as extra columns
        (the merge layer's concat) -> [R, O (+ d0)]."""
        lib = _native.load()
        dev = rres.device
        self._drop_calls = getattr(self, '_drop_calls', 0) + 1
        R, O, H, dh = rres.shape[0], self.out_dim, self.n_heads, self.head_dim
        d, D, T = self.node_dim, self.edge_dim, self.time_dim
        C = d + D + T
        WKV = self.W_KV.weight.detach()
        WK_t = WKV[:O].t().contiguous()  # [C, O]: B operand of the query fold
        WV = WKV[O:]  # [O, C]
        f32 = dict(dtype=torch.float32, device=dev)
        Q = torch.empty((R, O), **f32)
        _ops.sgemm_nt(rres, self.W_Q.weight.detach(), Q)
        # qf[r, h, :] = Q[r, head h] @ W_K[head h, :]      (one batched launch over heads)
        qf = torch.empty((R, H, C), **f32)
        _ops.sgemm_nt(Q, WK_t, qf, M=R, N=C, K=dh, batch=H, sA=dh, sB=dh, sC=C)
        zbar = torch.empty((R, H, C), **f32)
        stream = _native.stream_ptr()
        _native.check(
            lib.tgmx_tgat_attn_reduce(
                qf.data_ptr(), nbrf.data_ptr(), d, _native.ptr(ex), D, _native.ptr(seed_t), _native.ptr(nbr_t), _native.ptr(nbr_id),
                _native.ptr(tw), _native.ptr(tb), _native.ptr(nbr_time_feat), _native.ptr(mask), T, H, k, R, float(dh) ** -0.5, 0,
                zbar.data_ptr(), 0, self._dropout_site(0), stream,
            ),
            'tgmx_tgat_attn_reduce',
        )  # fmt: skip
        # Oattn[:, head h] = zbar[:, h, :] @ W_V[head h, :].T
        oattn = torch.empty((R, O), **f32)
        _ops.sgemm_nt(zbar.view(R, H * C), WV, oattn, M=R, N=dh, K=C, batch=H, sA=C, sB=dh * C, sC=dh)
        y = torch.empty((R, O), **f32)
        _ops.sgemm_nt(oattn, self.W_O.weight.detach(), y, bias=self.W_O.bias.detach())
        drop_y = self._dropout_site(1)
        if drop_y is not None:
            _native.check(lib.tgmx_dropout(y.data_ptr(), O, R, O, drop_y, y.data_ptr(), O, stream), 'tgmx_dropout')
        d0 = 0 if z0 is None else z0.shape[1]
        out = torch.empty((R, O + d0), **f32)
        ln = self.layer_norm
        _native.check(
            lib.tgmx_ln_residual_concat(y.data_ptr(), 0, rres.data_ptr(), 0, ln.weight.detach().data_ptr(), ln.bias.detach().data_ptr(), O,
                                        float(ln.eps), _native.ptr(z0), d0, R, out.data_ptr(), 0, stream),
            'tgmx_ln_residual_concat',
        )  # fmt: skip
        return out

    def forward(self, node_x: Tensor, time_feat: Tensor, edge_feat: Tensor, nbr_node_feat: Tensor, nbr_time_feat: Tensor,
                valid_nbr_mask: Tensor) -> Tensor:  # fmt: skip
        """Reference signature (attention.py:58-66): explicit time features and mask."""
        lib = _native.load()
        node_x = _ops._f32c(node_x, 'node_x')
        R, k = valid_nbr_mask.shape
        T, O = self.time_dim, self.out_dim
        rres = torch.empty((R, O), dtype=torch.float32, device=node_x.device)
        tf = _ops._f32c(time_feat, 'time_feat')
        _native.check(lib.tgmx_tgat_rres(node_x.data_ptr(), node_x.stride(0), self.node_dim, 0, tf.data_ptr(), T, O, R, rres.data_ptr(), 0, _native.stream_ptr()), 'tgmx_tgat_rres')
        mask = valid_nbr_mask.to(torch.uint8).contiguous()
        return self.attend(
            rres, _ops._f32c(nbr_node_feat, 'nbr_node_feat'), _ops._f32c(edge_feat, 'edge_feat'), k,
            nbr_time_feat=_ops._f32c(nbr_time_feat, 'nbr_time_feat'), mask=mask,
        )  # fmt: skip

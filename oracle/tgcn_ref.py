"""ORACLE (test infrastructure only -- never imported by the product path).

Plain torch restatement of the discrete-time cell of /root/reference/tgm/nn/encoder/tgcn.py:8-157:
    U = sigmoid(linear_u([conv_u(X) | H])),  R = sigmoid(linear_r([conv_r(X) | H])),
    C = tanh(linear_c([conv_c(X) | H * R])),  H' = U * H + (1 - U) * C
and of the third-party ``torch_geometric.nn.GCNConv`` (2.6.1) from its published definition
(gcn_norm: add_remaining_self_loops with fill 1 or 2, symmetric normalisation with the in-degree).

Parity status: the gate wiring is pinned against the reference's TGCN class (golden g10, recorded with
this same GCNConv restatement standing in for PyG); GCNConv itself is UNPINNED upstream (PyG cannot be
installed here; the reference's own tests check shapes only).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor


def gcn_conv_ref(x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor], W: Tensor, b: Tensor, improved: bool = False,
                 add_self_loops: bool = True) -> Tensor:  # fmt: skip
    N = x.shape[0]
    src, dst = edge_index[0].long(), edge_index[1].long()
    w = torch.ones(src.numel()) if edge_weight is None else edge_weight.float()
    if add_self_loops:
        fill = 2.0 if improved else 1.0
        loop = src == dst
        loop_w = torch.full((N,), fill)
        loop_w[dst[loop]] = w[loop]  # an existing self loop keeps its weight
        src = torch.cat([src[~loop], torch.arange(N)])
        dst = torch.cat([dst[~loop], torch.arange(N)])
        w = torch.cat([w[~loop], loop_w])
    deg = torch.zeros(N).index_add_(0, dst, w)
    dinv = deg.pow(-0.5)
    dinv[torch.isinf(dinv)] = 0
    norm = dinv[src] * w * dinv[dst]
    xw = x @ W.T
    out = torch.zeros(N, W.shape[0]).index_add_(0, dst, norm[:, None] * xw[src])
    return out + b


def tgcn_cell_ref(p: Dict[str, Tensor], x: Tensor, edge_index: Tensor, edge_weight: Optional[Tensor] = None, H: Optional[Tensor] = None,
                  improved: bool = False, add_self_loops: bool = True) -> Tensor:  # fmt: skip
    C = p['linear_u.weight'].shape[0]
    H = torch.zeros(x.shape[0], C) if H is None else H
    conv = lambda g: gcn_conv_ref(x, edge_index, edge_weight, p[f'conv_{g}.lin.weight'], p[f'conv_{g}.bias'], improved, add_self_loops)
    lin = lambda g, v: v @ p[f'linear_{g}.weight'].T + p[f'linear_{g}.bias']
    U = torch.sigmoid(lin('u', torch.cat([conv('u'), H], 1)))
    R = torch.sigmoid(lin('r', torch.cat([conv('r'), H], 1)))
    Cc = torch.tanh(lin('c', torch.cat([conv('c'), H * R], 1)))
    return U * H + (1 - U) * Cc

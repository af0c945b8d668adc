"""ORACLE (test infrastructure only -- never imported by the product path).

The *restructured* TGAT attention the HIP kernels implement, restated in plain
torch fp32 so that the algebra can be checked on CPU against the golden vectors
independently of any kernel bug (tests/test_tgat_fold_cpu.py).

The reference projects every neighbor slot through W_KV (attention.py:98-101):
2*R*k*C*2O flops, 28 GFLOP per batch at the headline config.  Because the query
length is 1, attention is linear in K and V, so the projection can be folded to
the query / output side:

    score[r,h,s] = (Q[r,h,:] . (W_K,h z[r,s,:])) * dh^-1/2 = (W_K,h^T Q[r,h,:]) . z[r,s,:] * dh^-1/2
    O[r,h,:]     = sum_s A[r,h,s] (W_V,h z[r,s,:])          = W_V,h (sum_s A[r,h,s] z[r,s,:])

i.e. per row one folded query qf[r,h,:] in R^C and one attention-weighted mean
zbar[r,h,:] in R^C; the per-slot work is two length-C dot/axpy passes over the
gathered features (HBM-bound, no GEMM), and the dense work shrinks to
[R, O] x [O, C]-sized contractions (~14x fewer flops).  Same mathematics, fp32
throughout; only the association of the sums differs.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F
from torch import Tensor

from .tgat_ref import PAD_ID, layer_dropout, merge, time2vec


def temporal_attention_folded(p: Dict[str, Tensor], prefix: str, n_heads: int, node_x, time_feat, edge_feat, nbr_node_feat, nbr_time_feat, mask, drop=None):
    WQ, WKV, WO, bO = p[prefix + 'W_Q.weight'], p[prefix + 'W_KV.weight'], p[prefix + 'W_O.weight'], p[prefix + 'W_O.bias']
    g, be = p[prefix + 'layer_norm.weight'], p[prefix + 'layer_norm.bias']
    O = WQ.shape[0]
    H, dh = n_heads, O // n_heads
    WK, WV = WKV[:O], WKV[O:]
    pad = O - node_x.shape[1] - time_feat.shape[1]
    X = F.pad(node_x, (0, pad)) if pad else node_x
    R = torch.cat([X, time_feat], dim=1)
    Q = R @ WQ.T  # [B, O]
    B, k = mask.shape
    C = WK.shape[1]
    # folded query per head: qf[b,h,:] = Q[b, head h] @ WK[head h, :]
    qf = torch.stack([Q[:, h * dh : (h + 1) * dh] @ WK[h * dh : (h + 1) * dh] for h in range(H)], dim=1)  # [B, H, C]
    Z = torch.cat([nbr_node_feat, edge_feat, nbr_time_feat], dim=-1)  # [B, k, C]  (never materialised on the GPU)
    A = torch.einsum('bhc,bkc->bhk', qf, Z) * dh**-0.5
    A = A.masked_fill(~mask[:, None, :], -1e10)
    A = torch.softmax(A, dim=-1)
    if drop is not None:
        A = A * drop[0].to(A.dtype)
    zbar = torch.einsum('bhk,bkc->bhc', A, Z)  # [B, H, C]
    Oattn = torch.cat([zbar[:, h] @ WV[h * dh : (h + 1) * dh].T for h in range(H)], dim=1)  # [B, O]
    out = Oattn @ WO.T + bO
    if drop is not None:
        out = out * drop[1].to(out.dtype)
    return F.layer_norm(out + R, (O,), g, be, 1e-5)


def tgat_forward_folded(p, n_heads, node_x, seed_nids, seed_times, nbr_nids, nbr_edge_x, nbr_edge_time, dropout=None) -> Tensor:
    L = len(nbr_nids)
    tw, tb = p['time_encoder.w.weight'], p['time_encoder.w.bias']
    z = {0: {0: node_x[seed_nids[0].long()]}}
    for i in range(1, L + 1):
        z[0][i] = node_x[nbr_nids[i - 1].reshape(-1).long()]
    for j in range(1, L + 1):
        z[j] = {}
        for i in range(L - j + 1):
            x = z[j - 1][i]
            n, k = x.shape[0], nbr_nids[j - 1].shape[-1]
            out = temporal_attention_folded(
                p, f'attn.{j - 1}.', n_heads, node_x=x, time_feat=time2vec(torch.zeros(n), tw, tb), edge_feat=nbr_edge_x[i],
                nbr_node_feat=z[j - 1][i + 1].reshape(n, k, -1),
                nbr_time_feat=time2vec(seed_times[i][:, None] - nbr_edge_time[i], tw, tb), mask=nbr_nids[i] != PAD_ID,
                drop=layer_dropout(dropout, j, i, [z[0][q].shape[0] for q in range(i)], n, n_heads, k, p[f'attn.{j - 1}.W_Q.weight'].shape[0]),
            )  # fmt: skip
            z[j][i] = merge(p, f'merge_layers.{j - 1}.', out, z[0][i])
    return z[L][0]

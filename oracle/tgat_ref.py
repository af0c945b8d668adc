"""ORACLE (test infrastructure only -- never imported by the product path).

Plain torch-fp32 restatement of the reference's TGAT forward (eval mode),
written from SURVEY.md Appendix C.  Floating-point path, so this is the "torch
fp32 reference" the GPU kernels are compared against (tolerance 1e-5 relative,
stated in the tests); it is itself pinned against the reference's outputs in
tests/golden/g5_*.npz and g6_time2vec.npz by tests/test_oracle_golden.py.

Follows /root/reference/tgm/nn:
  modules/time_encoding.py:22-24   Time2Vec: cos(Linear(1->T)(float32(t)))
  modules/attention.py:58-128      TemporalAttention.forward
  encoder/tgat.py:36-38, 95-149    MergeLayer, TGAT.forward (hop-tree recursion)

Parameters are passed as a plain ``dict[str, Tensor]`` with the reference's
``state_dict`` key names (time_encoder.w.weight, attn.0.W_Q.weight, ...).
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import Tensor

PAD_ID = -1


def time2vec(t: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """cos(float32(t) * w + b); ``w`` is the Linear(1, T) weight [T, 1], ``b`` its bias [T].

    The reference evaluates ``Linear(1, T)`` through the host BLAS, and whether ``x*w + b`` is
    rounded once (fused multiply-add) or twice depends on the CPU / BLAS kernel: in the build
    container (where the golden vectors were recorded) it is ONE rounding, on the MI355X host
    (128-core, different kernel selection) the same torch call rounds twice.  With timestamps
    ~1e6 that 1-ulp difference of the argument moves the cosine by up to ~0.1, so the oracle
    pins the arithmetic explicitly to what the goldens contain: the float32 product is exact in
    float64, the sum is rounded to float32 once."""
    x = t.unsqueeze(-1).float().double()
    arg = (x * w.reshape(-1).double() + b.double()).float()
    return torch.cos(arg)


def temporal_attention(p: Dict[str, Tensor], prefix: str, n_heads: int, node_x, time_feat, edge_feat, nbr_node_feat, nbr_time_feat, mask, drop=None):
    """``drop`` = (scale_attn [B, H, k], scale_out [B, O]): train-mode dropout with explicit masks (attention.py:119,126)."""
    WQ, WKV, WO, bO = p[prefix + 'W_Q.weight'], p[prefix + 'W_KV.weight'], p[prefix + 'W_O.weight'], p[prefix + 'W_O.bias']
    g, be = p[prefix + 'layer_norm.weight'], p[prefix + 'layer_norm.bias']
    O = WQ.shape[0]
    dh = O // n_heads
    pad = O - node_x.shape[1] - time_feat.shape[1]
    X = F.pad(node_x, (0, pad)) if pad else node_x
    R = torch.cat([X, time_feat], dim=1)  # [B, O] residual == query input
    Q = R @ WQ.T
    Z = torch.cat([nbr_node_feat, edge_feat, nbr_time_feat], dim=-1) @ WKV.T  # [B, k, 2O]
    K, V = Z[..., :O], Z[..., O:]
    B, k = mask.shape
    Qh = Q.view(B, n_heads, dh)
    Kh = K.view(B, k, n_heads, dh)
    Vh = V.view(B, k, n_heads, dh)
    A = torch.einsum('bhd,bkhd->bhk', Qh, Kh) * dh**-0.5
    A = A.masked_fill(~mask[:, None, :], -1e10)
    A = torch.softmax(A, dim=-1)
    if drop is not None:
        A = A * drop[0].to(A.dtype)
    Oattn = torch.einsum('bhk,bkhd->bhd', A, Vh).reshape(B, O)
    out = Oattn @ WO.T + bO
    if drop is not None:
        out = out * drop[1].to(out.dtype)
    return F.layer_norm(out + R, (O,), g, be, 1e-5)


def layer_dropout(dropout, j: int, level: int, rows_before, n: int, H: int, k: int, O: int):
    """The two dropout scales of layer j (1-based) for the rows of `level` (tgmx_tgat_model_t.drop: layer j uses stream
    call * 64 + 2 j for the attention weights and + 1 for the W_O output; a level's rows follow the earlier levels')."""
    if dropout is None or not dropout[0]:
        return None
    from .dropout_ref import dropout_scale

    p_, seed, call = dropout
    row0 = int(sum(rows_before))
    return dropout_scale(p_, seed, call * 64 + 2 * j, (n, H, k), row0), dropout_scale(p_, seed, call * 64 + 2 * j + 1, (n, O), row0)


def merge(p: Dict[str, Tensor], prefix: str, x1: Tensor, x2: Tensor) -> Tensor:
    h = torch.cat([x1, x2], dim=1) @ p[prefix + 'fc1.weight'].T + p[prefix + 'fc1.bias']
    return h.relu() @ p[prefix + 'fc2.weight'].T + p[prefix + 'fc2.bias']


def tgat_forward(
    p: Dict[str, Tensor],
    n_heads: int,
    node_x: Tensor,
    seed_nids: List[Tensor],
    seed_times: List[Tensor],
    nbr_nids: List[Tensor],
    nbr_edge_x: List[Tensor],
    nbr_edge_time: List[Tensor],
    dropout=None,
) -> Tensor:
    """``dropout`` = (p, seed, call): train mode with the kernels' counter-based masks (oracle/dropout_ref.py)."""
    L = len(nbr_nids)
    tw, tb = p['time_encoder.w.weight'], p['time_encoder.w.bias']
    # leaves: node_x[ids]; pad id -1 indexes the LAST row, exactly like the reference's fancy indexing
    z = {0: {0: node_x[seed_nids[0].long()]}}
    for i in range(1, L + 1):
        z[0][i] = node_x[nbr_nids[i - 1].reshape(-1).long()]
    for j in range(1, L + 1):
        z[j] = {}
        for i in range(L - j + 1):
            x = z[j - 1][i]
            n = x.shape[0]
            k = nbr_nids[j - 1].shape[-1]
            out = temporal_attention(
                p, f'attn.{j - 1}.', n_heads,
                node_x=x,
                time_feat=time2vec(torch.zeros(n), tw, tb),
                edge_feat=nbr_edge_x[i],
                nbr_node_feat=z[j - 1][i + 1].reshape(n, k, -1),
                nbr_time_feat=time2vec(seed_times[i][:, None] - nbr_edge_time[i], tw, tb),
                mask=nbr_nids[i] != PAD_ID,
                drop=layer_dropout(dropout, j, i, [z[0][q].shape[0] for q in range(i)], n, n_heads, k, p[f'attn.{j - 1}.W_Q.weight'].shape[0]),
            )  # fmt: skip
            z[j][i] = merge(p, f'merge_layers.{j - 1}.', out, z[0][i])
    return z[L][0]

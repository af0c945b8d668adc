"""``TGAT`` encoder (tgm/nn/encoder/tgat.py:41-149) on the HIP kernels of ``csrc/tgat.hip``.

Parameter names match the reference (``time_encoder.w.*``, ``attn.{l}.*``,
``merge_layers.{l}.fc{1,2}.*``) so checkpoints interchange.  Execution differs:

* every level of the hop tree that shares a layer's weights is processed as ONE row
  batch (layer 1 of a 2-layer model: the 600 seeds and their 12 000 hop-1 neighbors are
  12 600 rows of the same GEMMs), so the launch count does not grow with the tree;
* the per-slot key/value projection is folded away (see ``attention.py``), Time2Vec and
  the valid-neighbor mask are computed inside the attention kernel from the sampler's
  raw outputs (timestamps, ids): the [R, k, C] key tensor never exists.

Forward / eval only for now (see ``TemporalAttention``).
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn
from torch import Tensor

from .. import _native
from . import _ops
from .attention import TemporalAttention
from .time_encoding import Time2Vec


class MergeLayer(nn.Module):
    """fc2(relu(fc1([x1 | x2])))  (tgat.py:11-38)."""

    def __init__(self, in_dim1: int, in_dim2: int, hidden_dim: int, output_dim: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(in_dim1 + in_dim2, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, output_dim)

    def forward_cat(self, cat: Tensor, out: Tensor) -> Tensor:
        """``cat`` = [x1 | x2] already concatenated (the LayerNorm kernel writes it that way)."""
        h = torch.empty((cat.shape[0], self.fc1.out_features), dtype=torch.float32, device=cat.device)
        _ops.sgemm_nt(cat, self.fc1.weight.detach(), h, bias=self.fc1.bias.detach(), relu=True)
        return _ops.sgemm_nt(h, self.fc2.weight.detach(), out, bias=self.fc2.bias.detach())

    def forward(self, x1: Tensor, x2: Tensor) -> Tensor:
        cat = torch.cat([_ops._f32c(x1, 'x1'), _ops._f32c(x2, 'x2')], dim=1)
        out = torch.empty((cat.shape[0], self.fc2.out_features), dtype=torch.float32, device=cat.device)
        return self.forward_cat(cat, out)


class TGAT(nn.Module):
    def __init__(self, node_dim: int, edge_dim: int, time_dim: int, embed_dim: int, num_layers: int, n_heads: int = 2,
                 dropout: float = 0.1) -> None:  # fmt: skip
        super().__init__()
        self.num_layers, self.embed_dim, self.node_dim = num_layers, embed_dim, node_dim
        self.time_encoder = Time2Vec(time_dim=time_dim)
        self.attn, self.merge_layers = nn.ModuleList(), nn.ModuleList()
        for i in range(num_layers):
            self.attn.append(TemporalAttention(n_heads=n_heads, node_dim=node_dim if i == 0 else embed_dim, edge_dim=edge_dim,
                                               time_dim=time_dim, dropout=dropout))  # fmt: skip
            self.merge_layers.append(MergeLayer(in_dim1=self.attn[-1].out_dim, in_dim2=node_dim, hidden_dim=embed_dim, output_dim=embed_dim))

    def forward(self, node_x: Tensor, seed_nids: List[Tensor], seed_times: List[Tensor], nbr_nids: List[Tensor],
                nbr_edge_x: List[Tensor], nbr_edge_time: List[Tensor]) -> Tensor:  # fmt: skip
        """Same arguments as the reference (tgat.py:95-103): the sampler's per-hop lists.
        Returns the seeds' embeddings [len(seed_nids[0]), embed_dim]."""
        L = self.num_layers
        for m in self.attn:
            m._check_mode()
        lib = _native.load()
        node_x = _ops._f32c(node_x, 'node_x')
        dev = node_x.device
        d0 = node_x.shape[1]
        f32 = dict(dtype=torch.float32, device=dev)
        stream = _native.stream_ptr()
        tw = self.time_encoder.w.weight.detach().reshape(-1)
        tb = self.time_encoder.w.bias.detach()

        # level i of the hop tree: V_0 = seeds, V_i = hop-(i-1) neighbors flattened (pads included)
        ids = [seed_nids[0].contiguous()] + [nbr_nids[i].reshape(-1) for i in range(L)]
        sizes = [int(v.numel()) for v in ids]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        # leaves z0 for every level, one buffer: z0[offs[i]:offs[i+1]] = node_x[V_i] (pad -1 -> last row)
        z0 = torch.empty((offs[-1], d0), **f32)
        for i in range(L + 1):
            if sizes[i]:
                _ops.gather_rows(node_x, ids[i], out=z0[offs[i] : offs[i + 1]])

        prev = z0  # z^{j-1} for levels 0 .. L-j+1, rows laid out by `offs`
        for j in range(1, L + 1):
            attn, merge = self.attn[j - 1], self.merge_layers[j - 1]
            n_lvl = L - j + 1  # levels 0 .. L-j get a new embedding
            Rtot = offs[n_lvl]
            O, T, d = attn.out_dim, attn.time_dim, attn.node_dim
            k = nbr_nids[j - 1].shape[-1]
            rres = torch.empty((Rtot, O), **f32)
            _native.check(lib.tgmx_tgat_rres(prev.data_ptr(), prev.stride(0), d, tb.data_ptr(), 0, T, O, Rtot, rres.data_ptr(), stream), 'tgmx_tgat_rres')
            # per level the inputs live in different sampler tensors; everything dense runs once over Rtot rows
            H, dh, C = attn.n_heads, attn.head_dim, d + attn.edge_dim + T
            WKV = attn.W_KV.weight.detach()
            Q = torch.empty((Rtot, O), **f32)
            _ops.sgemm_nt(rres, attn.W_Q.weight.detach(), Q)
            qf = torch.empty((Rtot, H, C), **f32)
            _ops.sgemm_nt(Q, WKV[:O].t().contiguous(), qf, M=Rtot, N=C, K=dh, batch=H, sA=dh, sB=dh, sC=C)
            zbar = torch.empty((Rtot, H, C), **f32)
            for i in range(n_lvl):
                Ri = sizes[i]
                if not Ri:
                    continue
                if nbr_nids[i].shape[-1] != k:
                    raise ValueError('TGAT needs the same number of neighbors at every hop used by one layer')
                nbrf = prev[offs[i + 1] : offs[i + 1] + Ri * k]  # z^{j-1}_{i+1} viewed [Ri, k, d]
                ex = nbr_edge_x[i]
                ex = ex if ex.is_contiguous() else ex.contiguous()
                _native.check(
                    lib.tgmx_tgat_attn_reduce(
                        qf[offs[i]].data_ptr(), nbrf.data_ptr(), d, _native.ptr(ex), attn.edge_dim,
                        seed_times[i].contiguous().data_ptr(), nbr_edge_time[i].contiguous().data_ptr(),
                        nbr_nids[i].contiguous().data_ptr(), tw.data_ptr(), tb.data_ptr(), 0, 0, T, H, k, Ri, float(dh) ** -0.5,
                        zbar[offs[i]].data_ptr(), stream,
                    ),
                    'tgmx_tgat_attn_reduce',
                )  # fmt: skip
            oattn = torch.empty((Rtot, O), **f32)
            _ops.sgemm_nt(zbar.view(Rtot, H * C), WKV[O:], oattn, M=Rtot, N=dh, K=C, batch=H, sA=C, sB=dh * C, sC=dh)
            y = torch.empty((Rtot, O), **f32)
            _ops.sgemm_nt(oattn, attn.W_O.weight.detach(), y, bias=attn.W_O.bias.detach())
            cat = torch.empty((Rtot, O + d0), **f32)
            ln = attn.layer_norm
            _native.check(
                lib.tgmx_ln_residual_concat(y.data_ptr(), rres.data_ptr(), ln.weight.detach().data_ptr(), ln.bias.detach().data_ptr(), O,
                                            float(ln.eps), z0.data_ptr(), d0, Rtot, cat.data_ptr(), stream),
                'tgmx_ln_residual_concat',
            )  # fmt: skip
            nxt = torch.empty((Rtot, self.embed_dim), **f32)
            merge.forward_cat(cat, nxt)
            prev = nxt
        return prev[: sizes[0]]

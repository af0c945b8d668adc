"""``TGAT`` encoder (tgm/nn/encoder/tgat.py:41-149) on the HIP kernels of ``csrc/tgat.hip``.

Parameter names match the reference (``time_encoder.w.*``, ``attn.{l}.*``,
``merge_layers.{l}.fc{1,2}.*``) so checkpoints interchange.  Execution differs:

* every level of the hop tree that shares a layer's weights is processed as ONE row
  batch (layer 1 of a 2-layer model: the 600 seeds and their 12 000 hop-1 neighbors are
  12 600 rows of the same GEMMs), so the launch count does not grow with the tree;
* the per-slot key/value projection is folded away (see ``attention.py``), Time2Vec and
  the valid-neighbor mask are computed inside the attention kernel from the sampler's
  raw outputs (timestamps, ids): the [R, k, C] key tensor never exists.

With gradients enabled the forward keeps its intermediates and a hand-written backward
(``nn/_tgat_train.py``, ``csrc/tgat_bwd.hip``) produces the parameter gradients.  In ``.train()``
mode the reference's two dropout sites per layer (attention.py:119,126; default p = 0.1,
tgat.py:67) are active: counter-based masks (``tgmx_dropout_t``) seeded from
``torch.initial_seed()``, one stream per forward call, regenerated -- not stored -- by the backward.
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.nn as nn
from torch import Tensor

from .. import _native
from . import _ops
from ._paramver import TransientCaches, param_key, param_list
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


class TGAT(TransientCaches, nn.Module):
    _TRANSIENT = ('_desc_cache', '_desc_struct', '_fold_keep', '_workspace', '_last_hops')

    def __init__(self, node_dim: int, edge_dim: int, time_dim: int, embed_dim: int, num_layers: int, n_heads: int = 2,
                 dropout: float = 0.1) -> None:  # fmt: skip
        super().__init__()
        self.num_layers, self.embed_dim, self.node_dim = num_layers, embed_dim, node_dim
        # inference over the DISTINCT (id, time) rows of every level (tgmx_tgat_hop_t.seed_keyed; bit-identical embeddings).  Opt-in: at the
        # headline shape finding the distinct rows costs more than computing them (csrc/tgat.hip compact_wanted has the numbers); it pays
        # for batches of >= ~33 k rows per layer
        self.compact_rows = os.environ.get('TGMX_TGAT_COMPACT', '0') not in ('', '0')
        self.time_encoder = Time2Vec(time_dim=time_dim)
        self.attn, self.merge_layers = nn.ModuleList(), nn.ModuleList()
        for i in range(num_layers):
            self.attn.append(TemporalAttention(n_heads=n_heads, node_dim=node_dim if i == 0 else embed_dim, edge_dim=edge_dim,
                                               time_dim=time_dim, dropout=dropout))  # fmt: skip
            self.merge_layers.append(MergeLayer(in_dim1=self.attn[-1].out_dim, in_dim2=node_dim, hidden_dim=embed_dim, output_dim=embed_dim))

    # ------------------------------------------------------------------
    def _model_desc(self):
        """ctypes description of the parameters for ``tgmx_tgat_forward``.  The weights the kernels read in padded / transposed layouts
        live in persistent buffers that are refreshed by ONE native launch (``tgmx_pack2d``) whenever a parameter changed -- after
        every optimizer step when training (``_paramver.param_key``; the ~25 torch launches this took were 0.2 ms of every step) --
        and rebuilt only when a parameter was reallocated."""
        key = param_key(self)
        cached = getattr(self, '_desc_cache', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        if self.num_layers > _native.TGAT_MAX_LAYERS:
            raise NotImplementedError(f'tgm_amd TGAT supports up to {_native.TGAT_MAX_LAYERS} layers')
        skey = tuple((p.data_ptr(), p.dtype, p.is_contiguous(), tuple(p.shape)) for p in param_list(self))
        st = getattr(self, '_desc_struct', None)
        if st is None or st[0] != skey or getattr(self, '_desc_volatile', False):
            self._desc_volatile = False
            st = self._desc_struct = (skey,) + self._build_desc()
        _, m, keep, jobs = st
        for lo in range(0, len(jobs), _native.PACK_MAX_JOBS):
            part = jobs[lo : lo + _native.PACK_MAX_JOBS]
            arr = (_native.PackJob * len(part))(*part)
            _native.check(_native.load().tgmx_pack2d(arr, len(part), _native.stream_ptr()), 'tgmx_pack2d')
        self._desc_cache = (key, (m, keep))
        self._desc_folded = False
        return m, keep

    def _build_desc(self):
        """The model struct, the tensors it points into, and the repacking jobs that fill the derived ones."""
        m = _native.TgatModel()
        keep, jobs = [], []  # tensors the struct points into; tgmx_pack_job_t entries

        def f32(t: Tensor) -> Tensor:
            """the parameter itself (fp32, contiguous: the normal case) -- anything else is converted once per optimizer step by
            forcing a structural rebuild"""
            t = t.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
                self._desc_volatile = True  # a converted copy goes stale with the parameter: rebuild at the next change
            keep.append(t)
            return t

        def ptr(t: Tensor) -> int:
            return f32(t).data_ptr()

        def packed(src: Tensor, rows: int, cols: int, src_ld: int, dst: Tensor, dst_off: int, dst_rows: int, dst_cols: int, dst_ld: int,
                   transpose: bool = False, src_off: int = 0) -> None:  # fmt: skip
            jobs.append(_native.PackJob(src.data_ptr() + 4 * src_off, dst.data_ptr() + 4 * dst_off, src_ld, dst_ld, rows, cols, dst_rows, dst_cols,
                                        int(transpose), 0))  # fmt: skip

        def padded(t: Tensor, cols: int, row0: int = 0, rows: int = -1) -> int:
            """zero-padded copy [rows, cols] of rows [row0, row0 + rows) so that every row starts 16-byte aligned"""
            t = f32(t)
            rows = t.shape[0] - row0 if rows < 0 else rows
            out = torch.empty((rows, cols), dtype=torch.float32, device=t.device)
            keep.append(out)
            packed(t, rows, t.shape[1], t.shape[1], out, 0, rows, cols, cols, src_off=row0 * t.shape[1])
            return out.data_ptr()

        p4 = lambda x: (x + 3) // 4 * 4
        m.tw, m.tb = ptr(self.time_encoder.w.weight.reshape(-1)), ptr(self.time_encoder.w.bias)
        m.num_layers, m.d0 = self.num_layers, self.node_dim
        for l, (attn, merge) in enumerate(zip(self.attn, self.merge_layers)):
            ly = m.layers[l]
            O, H, dh = attn.out_dim, attn.n_heads, attn.head_dim
            C = attn.node_dim + attn.edge_dim + attn.time_dim
            WKV = f32(attn.W_KV.weight)  # [2 O, C]: keys, then values
            wkt = torch.empty((C, H * p4(dh)), dtype=torch.float32, device=WKV.device)  # W_K^T, heads p4(dh) apart
            keep.append(wkt)
            for h in range(H):  # dst[c, j] = W_K[h dh + j, c]
                packed(WKV, C, dh, C, wkt, h * p4(dh), C, p4(dh), H * p4(dh), transpose=True, src_off=h * dh * C)
            ly.W_Q, ly.W_K_t, ly.W_V = padded(attn.W_Q.weight, p4(O)), wkt.data_ptr(), padded(WKV, p4(C), row0=O, rows=O)
            ly.W_O, ly.b_O = padded(attn.W_O.weight, p4(O)), ptr(attn.W_O.bias)
            ly.ln_g, ly.ln_b, ly.ln_eps = ptr(attn.layer_norm.weight), ptr(attn.layer_norm.bias), float(attn.layer_norm.eps)
            ly.fc1_w, ly.fc1_b = padded(merge.fc1.weight, p4(O + self.node_dim)), ptr(merge.fc1.bias)
            ly.fc2_w, ly.fc2_b = padded(merge.fc2.weight, p4(merge.fc1.out_features)), ptr(merge.fc2.bias)
            ly.d, ly.D, ly.T, ly.O, ly.H = attn.node_dim, attn.edge_dim, attn.time_dim, O, H
            ly.emb, ly.emb_out = merge.fc1.out_features, merge.fc2.out_features
        return m, keep, jobs

    def _ensure_fold(self, m, keep) -> None:
        """Inference only: fold the query side of every layer onto the layer input (weights only; lives and dies
        with the parameter cache of ``_model_desc``).  M_h = W_K,h^T W_Q,h [C, O]; U_h = M_h[:, :d];
        v_h = M_h[:, O-T:] cos(tb) -- computed with the native GEMM."""
        if self._desc_folded:
            return
        keep = self._fold_keep = []  # (the struct's own tensors persist across refreshes; these are replaced with every fold)
        p4 = lambda x: (x + 3) // 4 * 4
        for l, attn in enumerate(self.attn):
            ly = m.layers[l]
            O, H, dh = attn.out_dim, attn.n_heads, attn.head_dim
            C = attn.node_dim + attn.edge_dim + attn.time_dim
            d_in, T_ = attn.node_dim, attn.time_dim
            Cp, dp = p4(C), p4(d_in)
            WKV = attn.W_KV.weight.detach().float()
            WQ = attn.W_Q.weight.detach().float()
            dev = WKV.device
            ct = _ops.time2vec(torch.zeros(1, device=dev), self.time_encoder.w.weight.detach().reshape(-1).float().contiguous(),
                               self.time_encoder.w.bias.detach().float().contiguous())  # [1, T] = cos(tb)
            U = torch.zeros((H * Cp, dp), dtype=torch.float32, device=dev)
            v = torch.zeros((H * Cp, 1), dtype=torch.float32, device=dev)
            for h in range(H):
                wk_t = WKV[h * dh : (h + 1) * dh].t().contiguous()  # [C, dh]
                wq_t = WQ[h * dh : (h + 1) * dh].t().contiguous()  # [O, dh]
                M = torch.empty((C, O), dtype=torch.float32, device=dev)
                _ops.sgemm_nt(wk_t, wq_t, M)
                U[h * Cp : h * Cp + C, :d_in] = M[:, :d_in]
                _ops.sgemm_nt(M[:, O - T_ :], ct, v[h * Cp : h * Cp + C])
            keep += [U, v]
            ly.qf_U, ly.qf_v = U.data_ptr(), v.data_ptr()
            ly.qf_lane = None
            D_ = attn.edge_dim
            if d_in == 1 and D_ > 0 and D_ % 4 == 0 and D_ // 4 <= 64 and T_ <= 128:
                # the same numbers in the order the attention kernel's lanes take them (tgmx_tgat_layer_t.qf_lane)
                tab = torch.zeros((H, 64, 16), dtype=torch.float32, device=dev)
                lanes = torch.arange(64, device=dev)
                vf, Uf = v.reshape(H, Cp), U[:, 0].reshape(H, Cp)

                def put(slot_v: int, slot_u: int, cols: Tensor, ok: Tensor) -> None:
                    tab[:, lanes[ok], slot_v] = vf[:, cols[ok]]
                    tab[:, lanes[ok], slot_u] = Uf[:, cols[ok]]

                for i in range(4):
                    put(i, 4 + i, d_in + 4 * lanes + i, 4 * lanes + i < D_)
                put(8, 9, d_in + D_ + lanes, lanes < T_)
                put(10, 11, d_in + D_ + lanes + 64, lanes + 64 < T_)
                put(12, 13, lanes, lanes < d_in)
                keep.append(tab)
                ly.qf_lane = tab.data_ptr()
            # the tail's weights in the tiled order its one-kernel form streams them in (tgmx_tgat_tile16)
            merge = self.merge_layers[l]
            lib = _native.load()

            def tiled(w: Tensor, heads: int = 1) -> int:
                w = w.detach().float().contiguous()
                n, k = w.shape[0] // heads, w.shape[1]
                per = lib.tgmx_tgat_tile16_floats(n, k)
                out = torch.empty(heads * per, dtype=torch.float32, device=dev)
                for h in range(heads):
                    _native.check(lib.tgmx_tgat_tile16(w[h * n :].data_ptr(), k, n, k, out[h * per :].data_ptr(), _native.stream_ptr()),
                                  'tgmx_tgat_tile16')
                keep.extend([w, out])
                return out.data_ptr()

            ly.W_V_t16, ly.W_O_t16 = tiled(WKV[O:], H), tiled(attn.W_O.weight)
            ly.W_V_t16c = None
            hb16 = (dh + 15) // 16 * 16
            if H == 2 and 2 * hb16 // 16 <= 12:  # both heads stacked, each zero-padded to whole 16-row blocks, tiled as one matrix
                stack = torch.zeros((2 * hb16, WKV.shape[1]), dtype=torch.float32, device=dev)
                for h in range(2):
                    stack[h * hb16 : h * hb16 + dh] = WKV[O + h * dh : O + (h + 1) * dh]
                ly.W_V_t16c = tiled(stack)
            ly.fc1_t16, ly.fc2_t16 = tiled(merge.fc1.weight), tiled(merge.fc2.weight)
        self._desc_folded = True

    def forward(self, node_x: Tensor, seed_nids: List[Tensor], seed_times: List[Tensor], nbr_nids: List[Tensor],
                nbr_edge_x: List[Tensor], nbr_edge_time: List[Tensor]) -> Tensor:  # fmt: skip
        """Same arguments as the reference (tgat.py:95-103): the sampler's per-hop lists.
        Returns the seeds' embeddings [len(seed_nids[0]), embed_dim].  One native call enqueues the
        whole forward (``tgmx_tgat_forward``)."""
        L = self.num_layers
        lib = _native.load()
        node_x = _ops._f32c(node_x, 'node_x')
        if node_x.shape[1] != self.node_dim:
            raise ValueError(f'node_x has {node_x.shape[1]} features, the model was built for {self.node_dim}')
        dev = node_x.device
        model, _keep = self._model_desc()
        hops = (_native.TgatHop * L)()
        hold = []
        c = lambda t: t if t.is_contiguous() else t.contiguous()
        seeds = c(seed_nids[0])
        S0 = seeds.numel()
        rows = S0
        from ..core.lazy import EdgeFeaturesById

        drop_p = float(self.attn[0].dropout.p) if self.training else 0.0
        saving = bool(S0) and (drop_p > 0 or (torch.is_grad_enabled() and any(p.requires_grad for p in param_list(self))))
        # edge features by id (RecencyNeighborHook(edge_features='by_id')): the attention kernels -- forward, and the backward inside
        # tgmx_tgat_backward -- read the rows of the resident store where they consume them; shapes they do not cover (and the backward
        # composed from Python, TGMX_TGAT_BWD=py) gather the rows first
        by_id = (isinstance(nbr_edge_x, EdgeFeaturesById) and not getattr(self, '_by_id_unsupported', False)
                 and not (saving and os.environ.get('TGMX_TGAT_BWD', '') == 'py'))
        # Hops sampled FOR one another by one sampler call (the lists carry its tag, every deeper hop's seeds ARE the hop above's
        # flattened outputs): slots with equal (id, time) are then the same row of every deeper level, and inference computes each
        # distinct row once (tgmx_tgat_hop_t.seed_keyed).  Anything else -- hand-made tensors, replaced items, tensors modified in
        # place -- gets the row-per-slot computation.
        tag = getattr(nbr_nids, 'tag', None)
        keyed = (self.compact_rows and L > 1 and tag is not None and getattr(nbr_edge_time, 'tag', None) is tag and getattr(seed_times, 'tag', None) is tag
                 and getattr(seed_nids, 'tag', None) is tag and getattr(nbr_edge_x, 'tag', None) is tag and tag.matches(nbr_nids, nbr_edge_time, nbr_edge_x)
                 and all(seed_nids[i].data_ptr() == nbr_nids[i - 1].data_ptr() and seed_times[i].data_ptr() == nbr_edge_time[i - 1].data_ptr()
                         and seed_nids[i].numel() == nbr_nids[i - 1].numel() == seed_times[i].numel() for i in range(1, L)))  # fmt: skip
        for i in range(L):
            nid, nt, st = c(nbr_nids[i]), c(nbr_edge_time[i]), c(seed_times[i])
            ex = None if by_id else c(nbr_edge_x[i])
            if nid.shape[0] != rows or st.numel() != rows:
                raise ValueError(f'hop {i}: expected {rows} rows, got nbr_nids {tuple(nid.shape)} / seed_times {tuple(st.shape)}')
            eid, table = (c(nbr_edge_x.eids[i]), nbr_edge_x.table) if by_id else (None, None)
            hold += [nid, nt, ex, st, eid, table]
            h = hops[i]
            h.seed_t, h.nbr_id, h.nbr_t, h.edge_x, h.k = st.data_ptr(), nid.data_ptr(), nt.data_ptr(), _native.ptr(ex), nid.shape[-1]
            h.nbr_eid, h.edge_table = _native.ptr(eid), _native.ptr(table)
            h.seed_keyed = 1 if (keyed and i >= 1) else 0
            rows *= nid.shape[-1]
        if saving:
            # training: same native forward with every intermediate kept, hand-written backward (nn/_tgat_train.py);
            # also the path that applies dropout (train mode under no_grad included, like the reference)
            from ._tgat_train import ByIdUnsupported, TGATFunction

            if any(float(a.dropout.p) != float(self.attn[0].dropout.p) for a in self.attn):
                raise NotImplementedError('tgm_amd TGAT: every attention layer must use the same dropout probability')
            self._drop_calls = getattr(self, '_drop_calls', 0) + 1
            if getattr(self, '_drop_seed', None) is None:
                self._drop_seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
            model.drop.p, model.drop.seed, model.drop.stream, model.drop.row0 = drop_p, self._drop_seed, self._drop_calls, 0

            flat = [hold[6 * i + j] for i in range(L) for j in (3, 0, 1, 2, 4, 5)]  # (seed_t, nbr_id, nbr_t, edge_x, eid, table) per hop
            try:
                return TGATFunction.apply(self, node_x, seeds, flat, [int(hops[i].k) for i in range(L)], *param_list(self))
            except ByIdUnsupported:  # a shape the by-id attention kernels do not cover: gather the rows, like everybody else
                self._by_id_unsupported = True
                return self.forward(node_x, seed_nids, seed_times, nbr_nids, nbr_edge_x, nbr_edge_time)
        out = torch.empty((S0, self.embed_dim), dtype=torch.float32, device=dev)
        if S0 == 0:
            return out
        self._last_hops = hops  # (tests / diagnostics: what the last inference call was given)
        self._ensure_fold(model, _keep)
        need = lib.tgmx_tgat_workspace_bytes(model, S0, hops)
        ws = getattr(self, '_workspace', None)
        if ws is None or ws.device != dev or ws.numel() < need:
            ws = self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        rc = lib.tgmx_tgat_forward(model, node_x.data_ptr(), node_x.shape[0], seeds.data_ptr(), S0, hops, ws.data_ptr(), ws.numel(), 0,
                                   out.data_ptr(), _native.stream_ptr())  # fmt: skip
        if rc == _native.E_UNSUPPORTED and by_id:
            # a shape the by-id attention kernel does not cover (n_heads > 2, k > 20, ...): gather the rows, like everybody else
            self._by_id_unsupported = True
            return self.forward(node_x, seed_nids, seed_times, nbr_nids, nbr_edge_x, nbr_edge_time)
        _native.check(rc, 'tgmx_tgat_forward')
        return out

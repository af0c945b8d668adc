"""Training path of ``TGAT``: a ``torch.autograd.Function`` whose forward is the same single native
call as inference (``tgmx_tgat_forward`` with ``save=1``: every layer keeps its intermediates and the
attention weights in the workspace) and whose backward is the hand-derived gradient of the folded
attention (``oracle/tgat_fold.py``), composed from the kernels of ``csrc/tgat_bwd.hip``:

    dW = A^T dY   -> ``tgmx_sgemm_tn``   (exact-fp32 MFMA, deterministic split reduction)
    dX = dY W     -> ``tgmx_sgemm_nt``   with transposed weight copies
    biases / LayerNorm / Time2Vec parameters -> ``tgmx_colsum``
    LayerNorm, ReLU, per-row attention backward -> dedicated kernels

Round 3: the composition runs inside ONE native call (``tgmx_tgat_backward``: same kernels, same order, bit-identical
gradients); the Python composition below stays as its readable specification and as the tests' cross-check
(``TGMX_TGAT_BWD=py``).

Gradients are produced for every parameter of the module (not for ``node_x`` or the sampled edge
features, which are data).  Dropout (train mode): the forward call's ``tgmx_dropout_t`` (p, seed,
stream) is kept on the context and the backward regenerates both masks of every layer from it.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
from torch import Tensor

from .. import _native
from ._paramver import param_list


def _p4(x: int) -> int:
    return (x + 3) // 4 * 4


class ByIdUnsupported(Exception):
    """the saving forward was handed edge features by id at a shape the by-id attention kernels do not cover"""


class TGATFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, node_x: Tensor, seeds: Tensor, hop_tensors: List[Tensor], ks: List[int], *params: Tensor) -> Tensor:
        lib = _native.load()
        L = module.num_layers
        dev = node_x.device
        model, keep = module._model_desc()
        hops = (_native.TgatHop * L)()
        for i in range(L):
            st, nid, nt, ex, eid, table = hop_tensors[6 * i : 6 * i + 6]
            h = hops[i]
            h.seed_t, h.nbr_id, h.nbr_t, h.edge_x, h.k = st.data_ptr(), nid.data_ptr(), nt.data_ptr(), _native.ptr(ex), ks[i]
            h.nbr_eid, h.edge_table = _native.ptr(eid), _native.ptr(table)  # edge features by id (ex is None then)
        S0 = seeds.numel()
        lay = _native.TgatLayout()
        _native.check(lib.tgmx_tgat_layout(model, S0, hops, 1, lay), 'tgmx_tgat_layout')
        ws = torch.empty(int(lay.total_bytes), dtype=torch.uint8, device=dev)
        out = torch.empty((S0, module.embed_dim), dtype=torch.float32, device=dev)
        rc = lib.tgmx_tgat_forward(model, node_x.data_ptr(), node_x.shape[0], seeds.data_ptr(), S0, hops, ws.data_ptr(), ws.numel(), 1,
                                   out.data_ptr(), _native.stream_ptr())  # fmt: skip
        if rc == _native.E_UNSUPPORTED and any(hop_tensors[6 * i + 4] is not None for i in range(L)):
            raise ByIdUnsupported()
        _native.check(rc, 'tgmx_tgat_forward')
        ctx.module, ctx.lay, ctx.ws, ctx.hop_tensors, ctx.ks, ctx.keep, ctx.S0 = module, lay, ws, hop_tensors, ks, keep, S0
        ctx.hops = hops  # (points into hop_tensors, kept above)
        ctx.n_params = len(params)
        ctx.drop = (float(model.drop.p), int(model.drop.seed), int(model.drop.stream))  # the model block is shared: copy
        return out

    @staticmethod
    def backward(ctx, dz: Tensor):
        composed = os.environ.get('TGMX_TGAT_BWD', '') == 'py'  # the launch-by-launch composition below (tests: same gradients)
        grads = (_backward if composed else _backward_native)(ctx, dz.contiguous().float())
        return (None, None, None, None, None) + tuple(grads)


_GRAD_FIELDS = {'W_Q.weight': 'W_Q', 'W_KV.weight': 'W_KV', 'W_O.weight': 'W_O', 'W_O.bias': 'b_O', 'layer_norm.weight': 'ln_g', 'layer_norm.bias': 'ln_b',
                'fc1.weight': 'fc1_w', 'fc1.bias': 'fc1_b', 'fc2.weight': 'fc2_w', 'fc2.bias': 'fc2_b'}  # fmt: skip


def _backward_native(ctx, dz: Tensor) -> List[Optional[Tensor]]:
    """One native call: ``tgmx_tgat_backward`` runs what :func:`_backward` composes, in the same order with the same kernels."""
    lib = _native.load()
    module = ctx.module
    dev = dz.device
    model, _keep = module._model_desc()  # the forward's (parameters unchanged since: the cached block)
    params = param_list(module)
    plan = module.__dict__.get('_tgmx_grad_plan')
    if plan is None or plan[0] is not params:  # names, padded sizes and struct fields: once per parameter set
        names = [n for n, _ in module.named_parameters()]
        sizes = [(p.numel() + 63) // 64 * 64 for p in params]
        fields = []
        for n in names:
            head, _, rest = n.partition('.')
            if head in ('attn', 'merge_layers'):
                idx, _, leaf = rest.partition('.')
                fields.append((int(idx), _GRAD_FIELDS[leaf]))
            else:
                fields.append((-1, 'tw' if n == 'time_encoder.w.weight' else 'tb'))
        plan = module.__dict__['_tgmx_grad_plan'] = (params, sizes, fields)
    _, sizes, fields = plan
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    views = [c[: p.numel()].view(p.shape) for c, p in zip(flat.split(sizes), params)]
    g = _native.TgatGrads()
    for v, (layer, field) in zip(views, fields):
        setattr(g if layer < 0 else g.layers[layer], field, v.data_ptr())
    drop_p, drop_seed, drop_stream = ctx.drop
    drop = _native.Dropout(float(drop_p), drop_seed & 0xFFFFFFFFFFFFFFFF, drop_stream & 0xFFFFFFFFFFFFFFFF, 0)
    need = int(lib.tgmx_tgat_backward_workspace_bytes(model, ctx.lay, ctx.hops))
    if need == 0:
        _native.check(-1, 'tgmx_tgat_backward_workspace_bytes')
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _native.check(
        lib.tgmx_tgat_backward(model, ctx.lay, ctx.hops, ctx.ws.data_ptr(), dz.data_ptr(), dz.stride(0), drop, g, ws.data_ptr(), need,
                               _native.stream_ptr()),
        'tgmx_tgat_backward',
    )  # fmt: skip
    return [v.to(p.dtype) if p.requires_grad else None for v, p in zip(views, params)]


def _backward(ctx, dz: Tensor) -> List[Tensor]:
    lib = _native.load()
    module, lay, ws, ks = ctx.module, ctx.lay, ctx.ws, ctx.ks
    drop_p, drop_seed, drop_stream = ctx.drop
    dev = dz.device
    L, d0 = module.num_layers, module.node_dim
    stream = _native.stream_ptr()
    f32 = dict(dtype=torch.float32, device=dev)
    base_off = (-ws.data_ptr()) % 256
    wsf = ws[base_off : base_off + (ws.numel() - base_off) // 4 * 4].view(torch.float32)

    def saved(off: int, rows: int, ld: int) -> Tensor:
        return wsf[off : off + rows * ld].view(rows, ld)

    def nt(A: Tensor, B: Tensor, C: Tensor, M: int, N: int, K: int, batch=1, sA=0, sB=0, sC=0) -> None:
        """C[M, N] = A[M, K] @ B[N, K]^T (row-major 2-D views; leading dims from the strides)"""
        _native.check(lib.tgmx_sgemm_nt(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0), M, N, K, 0, 0,
                                        batch, sA, sB, sC, stream), 'tgmx_sgemm_nt')  # fmt: skip

    tn_ws = [None]

    def tn(A: Tensor, B: Tensor, C: Tensor, R: int, M: int, N: int, batch=1, sA=0, sB=0, sC=0) -> None:
        """C[M, N] = A[:R, :M]^T @ B[:R, :N]"""
        need = lib.tgmx_sgemm_tn_workspace_bytes(R, M, N, batch)
        if tn_ws[0] is None or tn_ws[0].numel() * 4 < need:
            tn_ws[0] = torch.empty((need + 3) // 4, **f32)
        _native.check(lib.tgmx_sgemm_tn(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0), R, M, N, batch,
                                        sA, sB, sC, 0, tn_ws[0].data_ptr(), stream), 'tgmx_sgemm_tn')  # fmt: skip

    cs_ws = torch.empty(256 * 2048, **f32)

    def colsum(X: Tensor, R: int, C: int, out: Tensor, accumulate=False) -> None:
        assert C <= 2048
        _native.check(lib.tgmx_colsum(X.data_ptr(), X.stride(0), R, C, out.data_ptr(), 1 if accumulate else 0, cs_ws.data_ptr(), stream), 'tgmx_colsum')

    time_enc = module.time_encoder
    T = time_enc.time_dim
    tw = time_enc.w.weight.detach().reshape(-1).contiguous()
    tb = time_enc.w.bias.detach().contiguous()
    d_tw = torch.zeros(T, **f32)
    d_tb = torch.zeros(T, **f32)
    param_grads = {}
    level_off, level_rows = list(lay.level_off), list(lay.level_rows)
    z0 = saved(lay.z0, level_off[L + 1], d0)

    dout = dz  # gradient of the current layer's output rows [R_j, emb_out]
    for j in range(L, 0, -1):
        attn, merge = module.attn[j - 1], module.merge_layers[j - 1]
        lo = lay.layers[j - 1]
        R, Op, dhp, Cp, Kc, Ep = int(lo.R), lo.Op, lo.dhp, lo.Cp, lo.Kc, lo.Ep
        O, H, dh = attn.out_dim, attn.n_heads, attn.head_dim
        d, D = attn.node_dim, attn.edge_dim
        C = d + D + T
        emb, emb_out = merge.fc1.out_features, merge.fc2.out_features
        k = ks[j - 1]
        n_lvl = L - j + 1
        rres, oattn, y = saved(lo.rres, R, Op), saved(lo.oattn, R, Op), saved(lo.y, R, Op)
        Q, qf, zbar = saved(lo.Q, R, H * dhp), saved(lo.qf, R, H * Cp), saved(lo.zbar, R, H * Cp)
        cat, h1, probs = saved(lo.cat, R, Kc), saved(lo.h1, R, Ep), saved(lo.probs, R, H * k)
        prev = z0 if j == 1 else saved(lay.layers[j - 2].out, level_off[n_lvl + 1], d)

        WKV = attn.W_KV.weight.detach().float()
        W_Q, W_O = attn.W_Q.weight.detach().float(), attn.W_O.weight.detach().float()
        F1, F2 = merge.fc1.weight.detach().float(), merge.fc2.weight.detach().float()

        # ---- merge MLP ----
        g_F2, g_f2b = torch.empty_like(F2), torch.empty(emb_out, **f32)
        tn(dout, h1, g_F2, R, emb_out, emb)
        colsum(dout, R, emb_out, g_f2b)
        dh1 = torch.empty((R, Ep), **f32)
        nt(dout, F2.t().contiguous(), dh1, R, emb, emb_out)
        _native.check(lib.tgmx_relu_mask(dh1.data_ptr(), Ep, h1.data_ptr(), Ep, R, emb, stream), 'tgmx_relu_mask')
        g_F1, g_f1b = torch.empty_like(F1), torch.empty(emb, **f32)
        tn(dh1, cat, g_F1, R, emb, O + d0)
        colsum(dh1, R, emb, g_f1b)
        dcat = torch.empty((R, Kc), **f32)
        nt(dh1, F1.t().contiguous(), dcat, R, O + d0, emb)
        # ---- LayerNorm(y + rres) ----
        ln = attn.layer_norm
        du, dgx = torch.empty((R, Op), **f32), torch.empty((R, Op), **f32)
        _native.check(
            lib.tgmx_ln_backward(dcat.data_ptr(), Kc, y.data_ptr(), Op, rres.data_ptr(), Op, ln.weight.detach().data_ptr(), O, float(ln.eps), R,
                                 du.data_ptr(), Op, dgx.data_ptr(), Op, stream),
            'tgmx_ln_backward',
        )  # fmt: skip
        g_ln_w, g_ln_b = torch.empty(O, **f32), torch.empty(O, **f32)
        colsum(dgx, R, O, g_ln_w)
        colsum(dcat, R, O, g_ln_b)
        # ---- W_O (its output went through dropout, attention.py:126: the gradient takes the same mask; the residual
        # branch below keeps the unmasked du) ----
        du_y = du
        if drop_p > 0:
            du_y = torch.empty((R, Op), **f32)
            _native.check(lib.tgmx_dropout(du.data_ptr(), Op, R, O, _native.dropout_desc(drop_p, drop_seed, drop_stream * 64 + 2 * j + 1),
                                           du_y.data_ptr(), Op, stream), 'tgmx_dropout')  # fmt: skip
        g_WO, g_bO = torch.empty_like(W_O), torch.empty(O, **f32)
        tn(du_y, oattn, g_WO, R, O, O)
        colsum(du_y, R, O, g_bO)
        doattn = torch.empty((R, Op), **f32)
        nt(du_y, W_O.t().contiguous(), doattn, R, O, O)
        # ---- W_V fold: oattn[:, head h] = zbar[:, h, :] @ W_V[head h]^T ----
        g_WKV = torch.empty_like(WKV)
        g_WK, g_WV = g_WKV[:O], g_WKV[O:]
        tn(doattn, zbar, g_WV, R, dh, C, batch=H, sA=dh, sB=Cp, sC=dh * C)
        WV_t = torch.zeros((C, H * dhp), **f32)  # W_V^T with heads dhp apart: B operand of dzbar_h = doattn_h @ W_V[head h]
        for h in range(H):
            WV_t[:, h * dhp : h * dhp + dh] = WKV[O + h * dh : O + (h + 1) * dh].t()
        dzbar = torch.empty((R, H * Cp), **f32)
        nt(doattn, WV_t, dzbar, R, C, dh, batch=H, sA=dh, sB=dhp, sC=Cp)
        # ---- per-row attention backward, level by level ----
        dqf = torch.empty((R, H * Cp), **f32)
        dtime = torch.empty((R, 2 * T), **f32)
        need_dprev = j > 1
        dprev = torch.zeros((level_off[n_lvl + 1], d), **f32) if need_dprev else None
        hop_t = ctx.hop_tensors
        for i in range(n_lvl):
            Ri = level_rows[i]
            if not Ri:
                continue
            st_i, nid_i, nt_i, ex_i = hop_t[6 * i : 6 * i + 4]
            o = level_off[i]
            nbrf = prev[level_off[i + 1] : level_off[i + 1] + Ri * k]
            _native.check(
                lib.tgmx_tgat_attn_backward(
                    qf[o].data_ptr(), probs[o].data_ptr(), dzbar[o].data_ptr(), nbrf.data_ptr(), d, _native.ptr(ex_i), D, st_i.data_ptr(),
                    nt_i.data_ptr(), tw.data_ptr(), tb.data_ptr(), T, H, k, Ri, float(dh) ** -0.5, Cp, dqf[o].data_ptr(),
                    dprev[level_off[i + 1]].data_ptr() if need_dprev else 0, dtime[o].data_ptr(),
                    _native.dropout_desc(drop_p, drop_seed, drop_stream * 64 + 2 * j, row0=o), stream,
                ),
                'tgmx_tgat_attn_backward',
            )  # fmt: skip
        colsum(dtime, R, T, d_tw, accumulate=True)
        colsum(dtime[:, T:], R, T, d_tb, accumulate=True)
        # ---- W_K fold: qf[:, h, :] = Q[:, head h] @ W_K[head h] ----
        tn(Q, dqf, g_WK, R, dh, C, batch=H, sA=dhp, sB=Cp, sC=dh * C)
        dQ = torch.zeros((R, H * dhp), **f32)
        WK_p = torch.zeros((O, Cp), **f32)
        WK_p[:, :C] = WKV[:O]
        nt(dqf, WK_p, dQ, R, dh, C, batch=H, sA=Cp, sB=dh * Cp, sC=dhp)
        # ---- W_Q: Q[:, head h] = rres @ W_Q[head h rows]^T ----
        g_WQ = torch.empty_like(W_Q)
        tn(dQ, rres, g_WQ, R, dh, O, batch=H, sA=dhp, sB=0, sC=dh * O)
        WQ_t = torch.zeros((O, H * dhp), **f32)  # W_Q^T with the head blocks dhp apart, matching dQ's column layout
        for h in range(H):
            WQ_t[:, h * dhp : h * dhp + dh] = W_Q[h * dh : (h + 1) * dh].t()
        drres = torch.empty((R, Op), **f32)
        nt(dQ, WQ_t, drres, R, O, H * dhp)
        _native.check(lib.tgmx_add_cols(drres.data_ptr(), Op, du.data_ptr(), Op, R, O, 1, stream), 'tgmx_add_cols')  # + residual branch
        # rres = [x | 0 | cos(tb)]:  d tb -= sin(tb) * colsum(drres[:, time columns]);  d x = drres[:, :d]
        g_time_cols = torch.empty(T, **f32)
        colsum(drres[:, O - T :], R, T, g_time_cols)
        d_tb -= torch.sin(tb) * g_time_cols
        if need_dprev:
            _native.check(lib.tgmx_add_cols(dprev.data_ptr(), d, drres.data_ptr(), Op, R, d, 1, stream), 'tgmx_add_cols')

        pre = f'attn.{j - 1}.'
        param_grads[pre + 'W_Q.weight'], param_grads[pre + 'W_KV.weight'] = g_WQ, g_WKV
        param_grads[pre + 'W_O.weight'], param_grads[pre + 'W_O.bias'] = g_WO, g_bO
        param_grads[pre + 'layer_norm.weight'], param_grads[pre + 'layer_norm.bias'] = g_ln_w, g_ln_b
        mp = f'merge_layers.{j - 1}.'
        param_grads[mp + 'fc1.weight'], param_grads[mp + 'fc1.bias'] = g_F1, g_f1b
        param_grads[mp + 'fc2.weight'], param_grads[mp + 'fc2.bias'] = g_F2, g_f2b
        dout = dprev

    param_grads['time_encoder.w.weight'] = d_tw.view(T, 1)
    param_grads['time_encoder.w.bias'] = d_tb
    return [param_grads[name].to(p.dtype) if p.requires_grad else None for name, p in module.named_parameters()]

"""Training path of the TGN modules: ``torch.autograd.Function``s whose forwards are the same native kernels as
inference and whose backwards are hand-written (``csrc/tgn_bwd.hip``) or composed from the GEMM blocks of the TGAT
backward (``tgmx_sgemm_nt`` with transposed weights for dX, ``tgmx_sgemm_tn`` for dW, ``tgmx_colsum`` for biases and
the Time2Vec parameters).  The reference trains through torch autograd (``examples/linkproppred/tgn.py:97-118``);
``memory`` / ``last_update`` are buffers, so gradients reach the shared Time2Vec, the GRU cell and the TransformerConv
projections (and flow back through ``x``, the memory module's output).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from .. import _native
from . import _ops


def _colsum(X: Tensor, R: int, C: int) -> Tensor:
    lib = _native.load()
    out = torch.empty(C, dtype=torch.float32, device=X.device)
    ws = torch.empty(256 * C, dtype=torch.float32, device=X.device)
    _native.check(lib.tgmx_colsum(X.data_ptr(), X.stride(0), R, C, out.data_ptr(), 0, ws.data_ptr(), _native.stream_ptr()), 'tgmx_colsum')
    return out


def _sgemm_tn(A: Tensor, B: Tensor) -> Tensor:
    """A^T B: A [R, M], B [R, N] -> [M, N] (weight gradient: reduction over the rows)."""
    lib = _native.load()
    R, M, N = A.shape[0], A.shape[1], B.shape[1]
    C = torch.empty((M, N), dtype=torch.float32, device=A.device)
    need = int(lib.tgmx_sgemm_tn_workspace_bytes(R, M, N, 1))
    ws = torch.empty(max(need, 4), dtype=torch.uint8, device=A.device)
    _native.check(lib.tgmx_sgemm_tn(A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0), R, M, N, 1, 0, 0, 0, 0,
                                    ws.data_ptr(), _native.stream_ptr()), 'tgmx_sgemm_tn')  # fmt: skip
    return C


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) on the native GEMM."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
        x = x.contiguous()
        y = torch.empty((x.shape[0], weight.shape[0]), dtype=torch.float32, device=x.device)
        if x.shape[0]:
            _ops.sgemm_nt(x, weight.detach().contiguous(), y, bias=None if bias is None else bias.detach())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        R = x.shape[0]
        dx = dw = db = None
        if R == 0:
            return (torch.zeros_like(x) if ctx.needs_input_grad[0] else None, torch.zeros_like(weight),
                    torch.zeros(weight.shape[0], device=x.device) if ctx.has_bias else None)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _ops.sgemm_nt(dy, weight.detach().t().contiguous(), dx)  # dx = dy W
        if ctx.needs_input_grad[1]:
            dw = _sgemm_tn(dy, x)  # dW = dy^T x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum(dy, R, dy.shape[1])
        return dx, dw, db


class GruGateFn(torch.autograd.Function):
    """out = GRUCell gate arithmetic on precomputed gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh (h: buffer rows)."""

    @staticmethod
    def forward(ctx, gi: Tensor, gh: Tensor, h: Tensor) -> Tensor:
        lib = _native.load()
        R, M = h.shape
        out = torch.empty((R, M), dtype=torch.float32, device=h.device)
        _native.check(lib.tgmx_tgn_gru_gate(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), M, R, out.data_ptr(), _native.stream_ptr()), 'tgmx_tgn_gru_gate')
        ctx.save_for_backward(gi, gh, h)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        gi, gh, h = ctx.saved_tensors
        lib = _native.load()
        R, M = h.shape
        dgi, dgh = torch.empty_like(gi), torch.empty_like(gh)
        _native.check(lib.tgmx_tgn_gru_gate_backward(gi.data_ptr(), gh.data_ptr(), h.data_ptr(), dout.contiguous().data_ptr(), M, R,
                                                     dgi.data_ptr(), dgh.data_ptr(), _native.stream_ptr()), 'tgmx_tgn_gru_gate_backward')  # fmt: skip
        return dgi, dgh, None


class AggregateFn(torch.autograd.Function):
    """aggr rows of TGNMemory (tgmx_tgn_aggregate); differentiable w.r.t. the Time2Vec weight / bias."""

    @staticmethod
    def forward(ctx, tw: Tensor, tb: Tensor, mem_module, nodes: Tensor, assoc=None, stamp: int = 0):
        m = mem_module
        lib = _native.load()
        dev, R, M, D, T = nodes.device, nodes.numel(), m.memory_dim, m.raw_msg_dim, m.time_dim
        W = 2 * M + D + T
        aggr = torch.empty((R, W), dtype=torch.float32, device=dev)
        new_lu = torch.empty(R, dtype=torch.int64, device=dev)
        twc, tbc = tw.detach().reshape(-1).contiguous(), tb.detach().contiguous()
        _native.check(
            lib.tgmx_tgn_aggregate(
                nodes.data_ptr(), R, m.memory.data_ptr(), m.last_update.data_ptr(), M, m.num_nodes, m._st_lo[0].data_ptr(),
                m._st_cnt[0].data_ptr(), m._st_lo[1].data_ptr(), m._st_cnt[1].data_ptr(), _native.ptr(m._log_other), _native.ptr(m._log_t),
                _native.ptr(m._log_raw), D, twc.data_ptr(), tbc.data_ptr(), T, m.aggr_module.mean, aggr.data_ptr(), new_lu.data_ptr(),
                _native.ptr(assoc), stamp, _native.stream_ptr(),
            ),
            'tgmx_tgn_aggregate',
        )  # fmt: skip
        # the reference's training loop calls update_state BEFORE loss.backward(): snapshot the rows' message windows and
        # last_update now (the event log itself is append-only within an epoch; ctx keeps this log tensor alive)
        idx = nodes.long()
        idx = torch.where(idx < 0, idx + m.num_nodes, idx)
        ctx.rows = (m._st_lo[0][idx], m._st_cnt[0][idx], m._st_lo[1][idx], m._st_cnt[1][idx], m.last_update[idx], m._log_t)
        ctx.dims = (R, M, D, T, m.aggr_module.mean)
        ctx.twc, ctx.tbc, ctx.tw_shape = twc, tbc, tw.shape
        ctx.mark_non_differentiable(new_lu)
        return aggr, new_lu

    @staticmethod
    def backward(ctx, d_aggr: Tensor, _d_lu):
        lo0, c0, lo1, c1, lu, log_t = ctx.rows
        R, M, D, T, mean = ctx.dims
        lib = _native.load()
        dev = ctx.twc.device
        if R == 0:
            return torch.zeros(ctx.tw_shape, device=dev), torch.zeros(T, device=dev), None, None, None, None
        part = torch.empty((R, 2 * T), dtype=torch.float32, device=dev)
        _native.check(
            lib.tgmx_tgn_aggregate_backward(R, lo0.data_ptr(), c0.data_ptr(), lo1.data_ptr(), c1.data_ptr(), lu.data_ptr(), _native.ptr(log_t),
                                            M, D, ctx.twc.data_ptr(), ctx.tbc.data_ptr(), T, mean, d_aggr.contiguous().data_ptr(),
                                            part.data_ptr(), _native.stream_ptr()),
            'tgmx_tgn_aggregate_backward',
        )  # fmt: skip
        g = _colsum(part, R, 2 * T)
        return g[:T].reshape(ctx.tw_shape), g[T:], None, None, None, None


class EdgeAttrFn(torch.autograd.Function):
    """edge_attr = [Time2Vec(last_update[src] - t) | msg] (tgmx_tconv_edge_attr); differentiable w.r.t. Time2Vec."""

    @staticmethod
    def forward(ctx, tw: Tensor, tb: Tensor, lu_local: Tensor, src: Tensor, t: Tensor, msg: Tensor) -> Tensor:
        lib = _native.load()
        E, T, D = src.numel(), tb.numel(), msg.shape[1]
        out = torch.empty((E, T + D), dtype=torch.float32, device=msg.device)
        twc, tbc = tw.detach().reshape(-1).contiguous(), tb.detach().contiguous()
        _native.check(lib.tgmx_tconv_edge_attr(lu_local.data_ptr(), src.data_ptr(), t.data_ptr(), msg.data_ptr(), twc.data_ptr(), tbc.data_ptr(),
                                               T, D, E, out.data_ptr(), _native.stream_ptr()), 'tgmx_tconv_edge_attr')  # fmt: skip
        ctx.saved = (lu_local, src, t, twc, tbc, T, D, E, tw.shape)
        return out

    @staticmethod
    def backward(ctx, d_attr: Tensor):
        lu_local, src, t, twc, tbc, T, D, E, tw_shape = ctx.saved
        lib = _native.load()
        if E == 0:
            return torch.zeros(tw_shape, device=twc.device), torch.zeros(T, device=twc.device), None, None, None, None
        part = torch.empty((E, 2 * T), dtype=torch.float32, device=twc.device)
        _native.check(lib.tgmx_tconv_edge_attr_backward(lu_local.data_ptr(), src.data_ptr(), t.data_ptr(), twc.data_ptr(), tbc.data_ptr(),
                                                        d_attr.contiguous().data_ptr(), T, D, E, part.data_ptr(), _native.stream_ptr()),
                      'tgmx_tconv_edge_attr_backward')  # fmt: skip
        g = _colsum(part, E, 2 * T)
        return g[:T].reshape(tw_shape), g[T:], None, None, None, None


class TconvAttendFn(torch.autograd.Function):
    """out = skip + per-target softmax attention (tgmx_tconv_attend)."""

    @staticmethod
    def forward(ctx, q: Tensor, k: Tensor, v: Tensor, eproj: Tensor, skip: Tensor, order: Tensor, src: Tensor, seg_lo: Tensor,
                seg_hi: Tensor, H: int, C: int, drop: tuple = (0.0, 0, 0)) -> Tensor:  # fmt: skip
        lib = _native.load()
        out = skip.clone()
        U = q.shape[0]
        _native.check(lib.tgmx_tconv_attend(q.data_ptr(), k.data_ptr(), v.data_ptr(), eproj.data_ptr(), order.data_ptr(), src.data_ptr(),
                                            seg_lo.data_ptr(), seg_hi.data_ptr(), U, H, C, float(C) ** -0.5, out.data_ptr(),
                                            _native.dropout_desc(*drop), _native.stream_ptr()),
                      'tgmx_tconv_attend')  # fmt: skip
        ctx.save_for_backward(q, k, v, eproj, order, src, seg_lo, seg_hi)
        ctx.H, ctx.C, ctx.drop = H, C, drop
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        q, k, v, eproj, order, src, seg_lo, seg_hi = ctx.saved_tensors
        lib = _native.load()
        dout = dout.contiguous()
        dq = torch.empty_like(q)
        dk, dv = torch.zeros_like(k), torch.zeros_like(v)
        de = torch.empty_like(eproj)
        _native.check(lib.tgmx_tconv_attend_backward(q.data_ptr(), k.data_ptr(), v.data_ptr(), eproj.data_ptr(), order.data_ptr(), src.data_ptr(),
                                                     seg_lo.data_ptr(), seg_hi.data_ptr(), q.shape[0], ctx.H, ctx.C, float(ctx.C) ** -0.5,
                                                     dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), de.data_ptr(),
                                                     _native.dropout_desc(*ctx.drop), _native.stream_ptr()), 'tgmx_tconv_attend_backward')  # fmt: skip
        return dq, dk, dv, de, dout, None, None, None, None, None, None, None

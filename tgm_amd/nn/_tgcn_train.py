"""Training path of ``TGCN`` / ``GCNConv`` (tgm/nn/encoder/tgcn.py:8-157 under ``loss.backward()``, examples/nodeproppred/tgcn.py:92):
``torch.autograd.Function``s whose forward is the inference arithmetic (same kernels, same order: bit-identical outputs) and whose
backward is hand-written -- two element-wise kernels of ``csrc/tgcn.hip`` around the exact-fp32 MFMA GEMMs (``sgemm_nt`` with a
transposed weight for the data gradients, ``sgemm_tn`` for the weight gradients, ``colsum`` for the biases).  The normalised adjacency
carries no gradient (edge weights are data).  Gradients: every parameter, ``node_x`` and the recurrent state ``H``.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from .. import _native
from . import _ops
from ._tgn_train import _colsum, _sgemm_tn


class AdjMatmulFn(torch.autograd.Function):
    """out = A_hat y   (A_hat [N, N] dense, possibly a column view of a padded buffer; y [N, C]); d y = A_hat^T d out."""

    @staticmethod
    def forward(ctx, A: Tensor, y: Tensor) -> Tensor:
        N, C = y.shape
        out = torch.empty((N, C), dtype=torch.float32, device=y.device)
        _ops.sgemm_nt(A, y.detach().t().contiguous(), out, K=N)
        ctx.save_for_backward(A)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        (A,) = ctx.saved_tensors
        return None, _sgemm_tn(A, dout.contiguous())


class TGCNCellFn(torch.autograd.Function):
    """The whole cell.  Parameter order: conv_{u,r,c}.lin.weight, conv_{u,r,c}.bias, linear_{u,r,c}.weight, linear_{u,r,c}.bias."""

    @staticmethod
    def forward(ctx, x: Tensor, A: Tensor, H: Tensor, *params: Tensor) -> Tensor:
        lib = _native.load()
        stream = _native.stream_ptr()
        cw, cb, lw, lb = params[0:3], params[3:6], params[6:9], params[9:12]
        N, C, dev = x.shape[0], lw[0].shape[0], x.device
        f32 = dict(dtype=torch.float32, device=dev)
        W3 = torch.cat([w.detach() for w in cw])
        b3 = torch.cat([b.detach() for b in cb])
        xwt = torch.empty((3 * C, N), **f32)
        _ops.sgemm_nt(W3, x, xwt)
        G = torch.empty((N, 3 * C), **f32)
        _ops.sgemm_nt(A, xwt, G, bias=b3, K=N)
        cats = [torch.empty((N, 2 * C), **f32) for _ in range(3)]
        pre = [torch.empty((N, C), **f32) for _ in range(3)]  # u, r, c pre-activations
        for g in range(3):
            gate = pre[1] if g == 2 else None
            _native.check(lib.tgmx_tgcn_concat(G[:, g * C :].data_ptr(), 3 * C, H.data_ptr(), _native.ptr(gate), C, N, cats[g].data_ptr(), stream),
                          'tgmx_tgcn_concat')
            _ops.sgemm_nt(cats[g], lw[g].detach().contiguous(), pre[g], bias=lb[g].detach())
        out = torch.empty((N, C), **f32)
        _native.check(lib.tgmx_tgcn_output(pre[0].data_ptr(), pre[2].data_ptr(), H.data_ptr(), N * C, out.data_ptr(), stream), 'tgmx_tgcn_output')
        ctx.save_for_backward(x, A, H, W3, *cats, *pre, *lw)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        lib = _native.load()
        stream = _native.stream_ptr()
        x, A, H, W3 = ctx.saved_tensors[:4]
        cats, pre, lw = ctx.saved_tensors[4:7], ctx.saved_tensors[7:10], ctx.saved_tensors[10:13]
        N, C, dev = x.shape[0], pre[0].shape[1], x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dout = dout.contiguous()
        du, dc, dr, dH = (torch.empty((N, C), **f32) for _ in range(4))
        _native.check(lib.tgmx_tgcn_gate_backward(dout.data_ptr(), pre[0].data_ptr(), pre[2].data_ptr(), H.data_ptr(), N * C, du.data_ptr(),
                                                  dc.data_ptr(), dH.data_ptr(), stream), 'tgmx_tgcn_gate_backward')  # fmt: skip
        # gradient of every gate's input [conv_g(X) | H (R)]: d cat_g = d pre_g W_g
        dcat = [torch.empty((N, 2 * C), **f32) for _ in range(3)]
        wt = [w.detach().t().contiguous() for w in lw]  # [2C, C]
        _ops.sgemm_nt(du, wt[0], dcat[0])
        _ops.sgemm_nt(dc, wt[2], dcat[2])
        _native.check(lib.tgmx_tgcn_reset_backward(dcat[2].data_ptr(), dcat[0].data_ptr(), None, pre[1].data_ptr(), H.data_ptr(), C, N, dr.data_ptr(),
                                                   dH.data_ptr(), stream), 'tgmx_tgcn_reset_backward')  # fmt: skip
        _ops.sgemm_nt(dr, wt[1], dcat[1])
        _native.check(lib.tgmx_tgcn_reset_backward(None, None, dcat[1].data_ptr(), None, None, C, N, None, dH.data_ptr(), stream),
                      'tgmx_tgcn_reset_backward')
        dpre = (du, dr, dc)
        d_lw = [_sgemm_tn(dpre[g], cats[g]) for g in range(3)]
        d_lb = [_colsum(dpre[g], N, C) for g in range(3)]
        # G = A_hat (X W3^T) + b3:  dG = the left halves of dcat;  d(X W3^T) = A_hat^T dG;  dW3 = that^T X;  dX = that W3
        dG = torch.cat([d[:, :C] for d in dcat], dim=1)
        db3 = _colsum(dG, N, 3 * C)
        dY = _sgemm_tn(A, dG)  # [N, 3C]
        dW3 = _sgemm_tn(dY, x)  # [3C, in]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _ops.sgemm_nt(dY, W3.t().contiguous(), dx)
        d_cw = [dW3[g * C : (g + 1) * C] for g in range(3)]
        d_cb = [db3[g * C : (g + 1) * C] for g in range(3)]
        return (dx, None, dH if ctx.needs_input_grad[2] else None, *d_cw, *d_cb, *d_lw, *d_lb)

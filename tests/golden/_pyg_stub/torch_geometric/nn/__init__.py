import math

import torch


class _M(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


Linear = AntiSymmetricConv = TransformerConv = ChebConv = _M


class GCNConv(torch.nn.Module):
    """Restated from the published definition of torch_geometric.nn.GCNConv (gcn_norm with
    add_remaining_self_loops, symmetric normalisation by the in-degree); used only so that the
    reference's TGCN class can be executed to record golden g10 (gate wiring)."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, add_self_loops=True, **kw):
        super().__init__()
        self.improved, self.add_self_loops = improved, add_self_loops
        self.lin = torch.nn.Linear(in_channels, out_channels, bias=False)
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        bound = math.sqrt(6.0 / (in_channels + out_channels))
        torch.nn.init.uniform_(self.lin.weight, -bound, bound)

    def forward(self, x, edge_index, edge_weight=None):
        N = x.shape[0]
        src, dst = edge_index[0].long(), edge_index[1].long()
        w = torch.ones(src.numel()) if edge_weight is None else edge_weight.float()
        if self.add_self_loops:
            loop = src == dst
            loop_w = torch.full((N,), 2.0 if self.improved else 1.0)
            loop_w[dst[loop]] = w[loop]
            src = torch.cat([src[~loop], torch.arange(N)])
            dst = torch.cat([dst[~loop], torch.arange(N)])
            w = torch.cat([w[~loop], loop_w])
        deg = torch.zeros(N).index_add_(0, dst, w)
        dinv = deg.pow(-0.5)
        dinv[torch.isinf(dinv)] = 0
        xw = self.lin(x)
        out = torch.zeros(N, xw.shape[1]).index_add_(0, dst, (dinv[src] * w * dinv[dst])[:, None] * xw[src])
        return out + self.bias

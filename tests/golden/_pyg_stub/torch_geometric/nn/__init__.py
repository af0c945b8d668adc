import torch
class _M(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
GCNConv = Linear = AntiSymmetricConv = TransformerConv = ChebConv = _M

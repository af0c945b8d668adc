import torch
def scatter(src, index, dim=0, dim_size=None, reduce='sum'):
    assert dim == 0
    n = dim_size if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    shape = (n,) + tuple(src.shape[1:])
    if reduce in ('sum', 'add', 'mean'):
        out = torch.zeros(shape, dtype=src.dtype, device=src.device)
        out.index_add_(0, index, src)
        if reduce == 'mean':
            cnt = torch.zeros(n, dtype=src.dtype, device=src.device)
            cnt.index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype, device=src.device))
            cnt = cnt.clamp(min=1)
            out = out / cnt.view((-1,) + (1,) * (src.dim() - 1))
        return out
    if reduce == 'max':
        out = torch.zeros(shape, dtype=src.dtype, device=src.device)
        idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
        return out.scatter_reduce(0, idx, src, reduce='amax', include_self=False)
    raise NotImplementedError(reduce)

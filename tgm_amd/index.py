"""Static per-node temporal index (CSR) over the device-resident edge store.

For a chronological loader with known batch boundaries, the state of the
reference's per-node ring buffers at the start of a batch is a pure function of
a static index: node n's adjacency entries ordered by
``(batch_idx, time, role, eid)`` (role 0 = the node is the edge's source, 1 =
destination), cut at the batch's first event.  That makes every batch -- and
every seed inside a batch -- an independent unit of work, which is what lets
batches be sharded across GPUs with no exchange.

Layout in HBM: ``indptr[N+1]`` int64 and ``adj[M]`` 16-byte records
``{nbr:int32, eid:int32, ts:int64}`` (M = E for directed, 2E otherwise), so a
lane fetches one record with one dwordx4 load and a window of B records is one
contiguous B*16-byte read.

The build is one-off setup and runs in the native library (``tgmx_csr_build``:
closed-form canonical positions, one rocPRIM radix sort of ``(node, position)``
keys, a binary search per node for ``indptr``, record packing).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Union

import torch
from torch import Tensor

from . import _native


@dataclass
class TemporalCSR:
    indptr: Tensor  # [N+1] int64
    adj: Tensor  # [M, 2] int64 == M 16-byte records
    num_nodes: int
    num_edges: int
    directed: bool
    batch_starts: Tensor  # [nb] int64 (device): first edge index of every batch

    @property
    def device(self) -> torch.device:
        return self.indptr.device

    def records(self):
        """(nbr int32 [M], eid int32 [M], ts int64 [M]) views -- for tests / debugging."""
        as32 = self.adj.view(torch.int32).view(-1, 4)
        return as32[:, 0], as32[:, 1], self.adj[:, 1]


def build_csr(
    src: Tensor,
    dst: Tensor,
    ts: Tensor,
    num_nodes: int,
    batch_starts: Union[Tensor, Sequence[int], None] = None,
    batch_size: Optional[int] = None,
    first_edge: int = 0,
    directed: bool = False,
    order: str = 'batch',
) -> TemporalCSR:
    """Index the stream ``(src, dst, ts)`` (device tensors, time-sorted).

    Batch boundaries are given either explicitly (``batch_starts``: increasing
    edge indices, the first batch starts at ``batch_starts[0]``) or as a fixed
    ``batch_size`` counted from ``first_edge``.  Edges before the first boundary
    form one leading batch.  ``order='event'`` ignores the batch schedule and orders a node's entries
    by ``(eid, role)`` -- the candidate order of the uniform sampler.
    """
    _native.require_device(src, 'edge stream')
    dev = src.device
    E = int(src.numel())
    if order not in ('batch', 'event'):
        raise ValueError(f"order must be 'batch' or 'event', got {order!r}")
    if order == 'event':
        starts = torch.empty(0, dtype=torch.int64, device=dev)
    elif batch_starts is None:
        if batch_size is None or batch_size <= 0:
            raise ValueError('build_csr needs batch_starts or a positive batch_size')
        starts = torch.arange(first_edge, max(E, first_edge + 1), batch_size, device=dev, dtype=torch.int64)
    else:
        starts = torch.as_tensor(batch_starts, dtype=torch.int64, device=dev)
    ts = ts.to(torch.int64).contiguous()
    src, dst = src.contiguous(), dst.contiguous()
    M = E if directed else 2 * E
    indptr = torch.empty(num_nodes + 1, device=dev, dtype=torch.int64)
    adj = torch.empty((max(M, 1), 2), device=dev, dtype=torch.int64)
    lib = _native.load()
    ws_bytes = int(lib.tgmx_csr_build_workspace_bytes(E, num_nodes, 1 if directed else 0))
    if ws_bytes == 0:
        _native.check(-2, 'tgmx_csr_build_workspace_bytes')
    workspace = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        _native.check(
            lib.tgmx_csr_build(
                src.data_ptr(), dst.data_ptr(), ts.data_ptr(), E, num_nodes, starts.data_ptr() if starts.numel() else None,
                -1 if order == 'event' else int(starts.numel()),
                1 if directed else 0, indptr.data_ptr(), adj.data_ptr(), workspace.data_ptr(), ws_bytes, status.data_ptr(),
                _native.stream_ptr(dev.index),
            ),
            'tgmx_csr_build',
        )  # fmt: skip
    if int(status.item()):
        raise ValueError(f'Edge endpoints must satisfy 0 <= x < {num_nodes}')
    return TemporalCSR(indptr, adj[:M] if M else adj[:0], num_nodes, E, directed, starts)

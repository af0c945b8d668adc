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

The build is one-off setup.  It orders entries with device-side torch sorts
(plumbing) and packs the records with a HIP kernel; the per-batch path never
touches torch ops.
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
) -> TemporalCSR:
    """Index the stream ``(src, dst, ts)`` (device tensors, time-sorted).

    Batch boundaries are given either explicitly (``batch_starts``: increasing
    edge indices, the first batch starts at ``batch_starts[0]``) or as a fixed
    ``batch_size`` counted from ``first_edge``.  Edges before the first boundary
    form one leading batch.
    """
    _native.require_device(src, 'edge stream')
    dev = src.device
    E = int(src.numel())
    eids = torch.arange(E, device=dev, dtype=torch.int64)
    if batch_starts is None:
        if batch_size is None or batch_size <= 0:
            raise ValueError('build_csr needs batch_starts or a positive batch_size')
        starts = torch.arange(first_edge, max(E, first_edge + 1), batch_size, device=dev, dtype=torch.int64)
    else:
        starts = torch.as_tensor(batch_starts, dtype=torch.int64, device=dev)
    bidx = torch.searchsorted(starts, eids, right=True)  # 0 for edges before the first boundary

    ts = ts.to(torch.int64)
    if directed:
        canon = eids
        node_c = src.long()
    else:
        # runs of equal (batch, time) are contiguous in eid; inside a run the
        # source-role entries come first, then the destination-role entries.
        tspan = int(ts.max().item()) + 1 if E else 1
        key = bidx * tspan + ts
        run_lo = torch.searchsorted(key, key, right=False)
        run_hi = torch.searchsorted(key, key, right=True)
        pos_src = run_lo + eids  # 2*run_lo + (eid - run_lo)
        pos_dst = run_hi + eids  # 2*run_lo + (run_hi - run_lo) + (eid - run_lo)
        canon = torch.empty(2 * E, device=dev, dtype=torch.int64)
        canon[pos_src] = eids
        canon[pos_dst] = eids + E
        node_c = torch.cat([src, dst]).long()[canon]
    order = torch.sort(node_c, stable=True).indices
    perm = canon[order].contiguous()
    counts = torch.bincount(node_c, minlength=num_nodes)
    indptr = torch.zeros(num_nodes + 1, device=dev, dtype=torch.int64)
    indptr[1:] = torch.cumsum(counts, 0)

    M = int(perm.numel())
    adj = torch.empty((max(M, 1), 2), device=dev, dtype=torch.int64)
    lib = _native.load()
    _native.check(
        lib.tgmx_pack_adj(
            perm.data_ptr(), M, src.data_ptr(), dst.data_ptr(), ts.data_ptr(), E, adj.data_ptr(), _native.stream_ptr()
        ),
        'tgmx_pack_adj',
    )
    return TemporalCSR(indptr, adj[:M] if M else adj[:0], num_nodes, E, directed, starts)

"""Uniform temporal neighbor sampling behind the reference's ``NeighborSamplerHook`` surface.

Reference: ``tgm/hooks/neighbors/uniform.py:16-142`` over ``DGStorageArrayBackend.get_nbrs``
(``tgm/core/_storage/backends/array_backend.py:108-171``), which walks every edge before the batch in a Python loop
per hop (quadratic per epoch, its own comment says so).  Here the candidates of a node are a prefix of its entries
in a static per-node index ordered by ``(eid, role)`` (``tgmx_csr_build(num_batches=-1)``), found by a 64-ary wave
search, and the draw is a virtual Fisher-Yates shuffle inside the kernel (``tgmx_uniform_lookup_csr``).

Parity: rows with at most k candidates are bit-exact with the reference (all neighbors, event order, left aligned);
rows that are sampled are uniformly random k-subsets like the reference's ``random.sample``, but drawn from a
counter-based generator (Python's Mersenne Twister state cannot be reproduced on the device), so there parity is
distributional.  As in the reference, every occurrence of a node in one hop gets the same row.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from .. import _native
from ..core.batch import DGBatch
from ..core.graph import DGraph
from ..index import TemporalCSR, build_csr
from .base import SeedableHook, StatelessHook
from .recency import RecencyNeighborHook
from .registry import hook


@hook
class NeighborSamplerHook(StatelessHook, SeedableHook):
    """Load neighbors from the DGraph with uniform sampling.

    Args:
        num_nbrs: neighbors to sample at each hop.
        seed_nodes_keys / seed_times_keys: batch attributes naming hop-0 seeds / query times.
        directed: only aggregate src->dst interactions.
        id: suffix for the hook name and every produced attribute.
        seed: seed of the device generator (extension; the reference uses Python's global ``random``).
    """

    _cls_requires = {'edge_src', 'edge_dst', 'edge_time'}
    _cls_produces = {'seed_nids', 'seed_times', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x', 'seed_node_nbr_mask'}

    def __init__(self, num_nbrs: List[int], seed_nodes_keys: List[str], seed_times_keys: List[str], directed: bool = False,
                 id: Optional[str] = None, seed: Optional[int] = None, validate: str = 'sync') -> None:  # fmt: skip
        super().__init__()
        if not len(num_nbrs):
            raise ValueError('num_nbrs must be non-empty')
        if not all(isinstance(x, int) and x > 0 for x in num_nbrs):
            raise ValueError('Each value in num_nbrs must be a positive integer')
        if len(seed_nodes_keys) != len(seed_times_keys):
            raise ValueError(
                f'len(seed_nodes_keys) ({len(seed_nodes_keys)}) != len(seed_times_keys) ({len(seed_times_keys)})\n'
                f'seed_nodes_keys={seed_nodes_keys}, seed_times_keys={seed_times_keys}'
            )
        if max(num_nbrs) > 64:
            raise NotImplementedError('tgm_amd NeighborSamplerHook supports at most 64 neighbors per hop')
        self._num_nbrs = list(num_nbrs)
        self._directed = bool(directed)
        self._seed_nodes_keys = list(seed_nodes_keys)
        self._seed_times_keys = list(seed_times_keys)
        self._warned_seed_None = False
        # validate (extension, like RecencyNeighborHook's): 'sync' = the reference's raise-per-call (one device -> host read per batch, which
        # makes every batch wait for its own kernels); 'deferred' = the seeds are still checked on the device, the error surfaces at check()
        if validate not in ('sync', 'deferred'):
            raise ValueError(f"validate must be 'sync' or 'deferred', got {validate!r}")
        self._validate = validate
        self._num_nodes = 0  # taken from the graph at the first call
        self._mask_cache: Dict[tuple, Dict[str, Tensor]] = {}
        self._rng_seed = int(seed) if seed is not None else int(torch.seed() & 0x7FFFFFFFFFFFFFFF)
        self._calls = 0
        self._csr: Optional[TemporalCSR] = None
        self._csr_store = None
        self._edge_time_np: Optional[np.ndarray] = None
        self._status: Optional[Tensor] = None
        self._id = id
        self.seed_keys = list(seed_nodes_keys)
        self.__post_init__()

    @property
    def num_nbrs(self) -> List[int]:
        return self._num_nbrs

    def _ensure_index(self, dg: DGraph, device: torch.device) -> TemporalCSR:
        store = dg._storage
        if self._csr is None or self._csr_store is not store or self._csr.device != device:
            if device.type != 'cuda':
                raise _native.NativeLibraryError(
                    f'NeighborSamplerHook got a batch on {device}; tgm_amd kernels run only on a ROCm device (no CPU fallback).'
                )
            arr = store.on(device)
            self._num_nodes = int(store.num_nodes_global)
            self._csr = build_csr(arr.src, arr.dst, arr.ts, self._num_nodes, order='event', directed=self._directed)
            self._csr_store = store
            self._edge_time_np = store._time_np[store._edge_pos_np]
            self._status = torch.zeros(1, dtype=torch.int32, device=device)
            self._mask_cache.clear()
        return self._csr

    def check(self) -> None:
        """Raise the ``ValueError`` the reference would have raised for bad seeds seen since the last check (one device -> host read)."""
        if self._status is not None and int(self._status.item()):
            self._status.zero_()
            raise ValueError(f'Seed nodes must satisfy 0 <= x < {self._num_nodes}')

    def __call__(self, dg: DGraph, batch: DGBatch) -> DGBatch:
        device = batch.edge_src.device
        L = len(self._num_nbrs)
        if self._csr is None or self._csr_store is not dg._storage or self._csr.device != device:
            self._ensure_index(dg, device)  # (a host-resident graph raises here: no CPU fallback, and no misleading seed-range error)
        seeds, seed_times, seed_mask = RecencyNeighborHook._get_seed_tensors(self, batch, device)
        D = dg.edge_x_dim or 0
        out_seed_n, out_seed_t, out_n, out_t, out_x = [], [], [], [], []
        if not seeds.numel():
            # reference: CPU empties for every hop (uniform.py:92-106)
            for _ in range(L):
                out_seed_n.append(torch.empty(0, dtype=torch.int32))
                out_seed_t.append(torch.empty(0, dtype=torch.int64))
                out_n.append(torch.empty(0, dtype=torch.int32))
                out_t.append(torch.empty(0, dtype=torch.int64))
                out_x.append(torch.empty(0, D, dtype=torch.float32))
        else:
            csr = self._ensure_index(dg, device)
            lib = _native.load()
            # "strictly before this batch": edges with time <= min(batch.edge_time) - 1 (uniform.py:118-123)
            n_edges = batch.edge_src.shape[0]
            if n_edges == 0:
                raise ValueError('NeighborSamplerHook needs a batch with at least one edge (the reference takes edge_time.min())')
            if batch._edge_lo is not None:
                tmin = int(self._edge_time_np[int(batch._edge_lo)])  # the store is time-sorted: the first edge is the earliest
            else:
                tmin = int(batch.edge_time.min())
            ev_hi = int(np.searchsorted(self._edge_time_np, tmin, side='left'))
            edge_x = dg._storage.on(device).edge_x
            self._calls += 1
            with torch.cuda.device(device):
                stream = _native.stream_ptr(device.index)
                cur_n, cur_t = seeds, seed_times
                for hop, k in enumerate(self._num_nbrs):
                    S = cur_n.numel()
                    nid = torch.empty((S, k), dtype=torch.int32, device=device)
                    nts = torch.empty((S, k), dtype=torch.int64, device=device)
                    nx = torch.empty((S, k, D), dtype=torch.float32, device=device)
                    _native.check(
                        lib.tgmx_uniform_lookup_csr(
                            csr.indptr.data_ptr(), csr.adj.data_ptr(), _native.ptr(edge_x), D, cur_n.data_ptr(), S, k, ev_hi,
                            self._num_nodes, 1 if hop else 0, self._rng_seed, (self._calls << 8) | hop, nid.data_ptr(), nts.data_ptr(),
                            nx.data_ptr(), self._status.data_ptr(), stream,
                        ),
                        'tgmx_uniform_lookup_csr',
                    )  # fmt: skip
                    out_seed_n.append(cur_n)
                    out_seed_t.append(cur_t)
                    out_n.append(nid)
                    out_t.append(nts)
                    out_x.append(nx)
                    cur_n, cur_t = nid.view(-1), nts.view(-1)
            if self._validate == 'sync':
                self.check()
        self.add_batch_attribute(batch, 'seed_nids', out_seed_n)
        self.add_batch_attribute(batch, 'seed_times', out_seed_t)
        self.add_batch_attribute(batch, 'nbr_nids', out_n)
        self.add_batch_attribute(batch, 'nbr_edge_time', out_t)
        self.add_batch_attribute(batch, 'nbr_edge_x', out_x)
        self.add_batch_attribute(batch, 'seed_node_nbr_mask', seed_mask)
        return batch

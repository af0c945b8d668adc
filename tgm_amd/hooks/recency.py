"""``RecencyNeighborHook`` -- k-most-recent temporal neighbor sampling on MI355X.

Drop-in for tgm/hooks/neighbors/recency.py:18 (same constructor, same
``requires`` / ``produces``, same six batch attributes with the same dtypes,
shapes, padding and ordering) whose per-batch work runs in the HIP kernels of
``csrc/recency.hip``:

``mode='ring'`` (default)
    The reference's state machine, kept on the device: per-node rings of
    ``B = max(num_nbrs)`` 16-byte records + their feature rows, looked up for
    every hop and then appended to with the batch's edges.  Works for any batch
    sequence, exactly like the reference.
``mode='csr'``
    Stateless lookup into a static per-node index of the resident stream
    (``tgm_amd.index``); valid for chronological loaders whose batch boundaries
    are known (``batch_size=`` / ``batch_starts=``).  No per-batch update, any
    batch can be sampled at any time on any GPU.

Seed validation (recency.py:173-237) happens inside the lookup kernel (device
status word).  ``validate='sync'`` (default) reads the word every call and
raises ``ValueError`` like the reference; ``'deferred'`` only raises when
:meth:`check` is called, keeping the stream fully asynchronous.
"""
from __future__ import annotations

import logging
import warnings
from typing import Dict, List, Optional, Sequence

import torch
from torch import Tensor

from .. import _native
from ..core import DGBatch, DGraph
from ..index import TemporalCSR, build_csr
from .base import SeedableHook, StatefulHook
from .registry import hook

logger = logging.getLogger('tgm_amd.hooks.recency')

_ST_SEED_RANGE, _ST_SEED_TIME, _ST_EDGE_RANGE, _ST_SCRATCH, _ST_TS_BOUND = 1, 2, 4, 8, 16


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _NullCtx()


def _on_device(device: torch.device):
    """Device guard that costs nothing when ``device`` is already current."""
    idx = device.index
    cur = _native._cuda_get_device  # the C accessor: torch.cuda.current_device() walks through _lazy_init (~2 us)
    if idx is None or idx == (cur() if cur is not None else torch.cuda.current_device()):
        return _NULL
    return torch.cuda.device(device)


@hook
class RecencyNeighborHook(StatefulHook, SeedableHook):
    """Load neighbors using recency sampling: each node keeps its most recent neighbors.

    Args:
        num_nodes: total number of nodes to track.
        num_nbrs: neighbors to sample at each hop.
        seed_nodes_keys / seed_times_keys: batch attributes naming hop-0 seeds / query times.
        directed: only aggregate src->dst interactions.
        id: suffix for the hook name and every produced attribute.
        mode: 'ring' (streaming state, default) or 'csr' (static index, stateless).
        validate: 'sync' | 'deferred' | 'off'.
        batch_size / batch_starts: loader schedule, required by ``mode='csr'``.
        edge_features: 'dense' (default, the reference's [S, k, D] copies) or 'by_id' (edge ids + lazy gather, see
            ``tgm_amd.core.lazy.EdgeFeaturesById``; batches must come from the graph store).
        adj_features: mode='csr' only (default on): a second copy of the edge features in the index's adjacency order.
        key_arith: ring mode only.  'int32' (default) reproduces the reference bit for bit,
            including the int32 wrap of its update sort key (recency.py:347) that leaves
            stale / empty ring slots at dataset scale; 'int64' is the intended per-node
            chronological order (what ``mode='csr'`` implements).

    Key words: k-hop neighbour, recency, historical.
    """

    _cls_requires = {'edge_src', 'edge_dst', 'edge_time'}
    _cls_produces = {'seed_nids', 'seed_times', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x', 'seed_node_nbr_mask'}

    def __init__(
        self,
        num_nodes: int,
        num_nbrs: List[int],
        seed_nodes_keys: List[str],
        seed_times_keys: List[str],
        directed: bool = False,
        id: Optional[str] = None,
        mode: str = 'ring',
        validate: str = 'sync',
        batch_size: Optional[int] = None,
        batch_starts: Optional[Sequence[int]] = None,
        key_arith: str = 'int32',
        adj_features: bool = True,
        edge_features: str = 'dense',
    ) -> None:
        super().__init__()
        if key_arith not in ('int32', 'int64'):
            raise ValueError(f"key_arith must be 'int32' or 'int64', got {key_arith!r}")
        self._key_wrap32 = 1 if key_arith == 'int32' else 0
        # mode='csr': keep a copy of the feature rows in adjacency order (False: gather by edge id from the store's edge_x)
        if edge_features not in ('dense', 'by_id'):
            raise ValueError(f"edge_features must be 'dense' or 'by_id', got {edge_features!r}")
        # 'by_id' (ours): publish the edge id behind every sampled slot instead of copying its feature row; batch.nbr_edge_x is an
        # EdgeFeaturesById (rows gathered from the resident store when -- and only if -- somebody indexes it)
        self._by_id = edge_features == 'by_id'
        self._adj_features = bool(adj_features)
        self._adj_features_max_bytes = 32 << 30
        if not len(num_nbrs):
            raise ValueError('num_nbrs must be non-empty')
        if not all(isinstance(x, int) and x > 0 for x in num_nbrs):
            raise ValueError('Each value in num_nbrs must be a positive integer')
        if len(seed_nodes_keys) != len(seed_times_keys):
            raise ValueError(
                f'len(seed_nodes_keys) ({len(seed_nodes_keys)}) != len(seed_times_keys) ({len(seed_times_keys)})\n'
                f'seed_nodes_keys={seed_nodes_keys}, seed_times_keys={seed_times_keys}'
            )
        if mode not in ('ring', 'csr'):
            raise ValueError(f"mode must be 'ring' or 'csr', got {mode!r}")
        if validate not in ('sync', 'deferred', 'off'):
            raise ValueError(f"validate must be 'sync', 'deferred' or 'off', got {validate!r}")
        if mode == 'csr' and batch_size is None and batch_starts is None:
            raise ValueError("mode='csr' needs the loader schedule: pass batch_size= or batch_starts=")

        self._num_nodes = int(num_nodes)
        self._num_nbrs = list(num_nbrs)
        self._max_nbrs = max(num_nbrs)
        self._directed = bool(directed)
        self._seed_nodes_keys = list(seed_nodes_keys)
        self._seed_times_keys = list(seed_times_keys)
        self._mode = mode
        self._validate = validate
        self._batch_size = batch_size
        self._batch_starts = batch_starts
        self._warned_seed_None = False
        self._warned_order = False
        self._last_batch_t = -(1 << 62)

        self._device: Optional[torch.device] = None
        self._edge_x_dim: Optional[int] = None
        # ring state (device)
        self._ring: Optional[Tensor] = None  # [N*B, 2] int64 == 16-byte records
        self._ring_x: Optional[Tensor] = None  # [N*B, D] float32
        self._write_pos: Optional[Tensor] = None  # [N] int32
        self._scratch: Optional[Tensor] = None
        self._scratch_edges = 0  # batch size the scratch was sized for
        self._status: Optional[Tensor] = None  # [1] int32 device status word
        # csr state
        self._csr: Optional[TemporalCSR] = None
        self._csr_store = None
        self._csr_first = 0
        self._csr_bounds = None
        self._epoch_lo: Optional[int] = None  # first edge index visible in this epoch

        # optional kernel timing (bench.py): every `profile_every`-th call brackets the
        # lookup launch of hop `profile_hop` with HIP events on the launch stream
        self._step: Optional[_native.RecencyStep] = None  # argument block of tgmx_recency_step (static fields pre-filled)
        self.profile_hop: Optional[int] = None
        self.profile_every: int = 1
        self.profile_log: List[tuple] = []
        self.profile_pool: List[_native.KernelTimer] = []  # pre-created timers, one consumed per timed launch
        self._calls = 0

        self._id = id
        self.seed_keys = list(seed_nodes_keys)
        self.__post_init__()

    # ------------------------------------------------------------------
    @property
    def num_nbrs(self) -> List[int]:
        return self._num_nbrs

    @property
    def mode(self) -> str:
        return self._mode

    def reset_state(self) -> None:
        """Forget all history (recency.py:111-117)."""
        self._last_batch_t = -(1 << 62)
        self._warned_order = False
        if self._mode == 'ring':
            if self._ring is not None:
                lib = _native.load()
                with torch.cuda.device(self._device):
                    _native.check(
                        lib.tgmx_ring_reset(
                            self._ring.data_ptr(), self._write_pos.data_ptr(), self._max_nbrs, self._num_nodes, _native.stream_ptr()
                        ),
                        'tgmx_ring_reset',
                    )
        else:
            self._epoch_lo = None  # re-anchored at the next batch's first edge

    @property
    def batches_are_independent(self) -> bool:
        """Over the static index a batch's result is a function of (index, batch start, epoch start) alone: a loader may
        hand this hook any subset of the schedule's batches (``DGDataLoader(batch_shard=(rank, world))``).  Streaming rings: no."""
        return self._mode == 'csr'

    def _anchor_epoch(self, first_edge: int) -> None:
        """The epoch starts at edge ``first_edge`` unless it has started already (a batch-sharded loader: this rank's first batch is not
        the schedule's first)."""
        if self._mode == 'csr' and self._epoch_lo is None:
            self._epoch_lo = int(first_edge)

    def _refresh_ts_bound(self, dg: DGraph) -> None:
        """The store is time-sorted and keeps a host copy of the timestamps: [0, last] bounds every batch of this graph
        (lets the large-batch update sort only the key bits that can be set); unknown / negative times: no promise."""
        store = getattr(dg, '_storage', None)  # a foreign DGraph (the reference's) may keep its store elsewhere: no promise then
        self._bound_store = store
        bound = 0
        times = getattr(store, '_time_np', None)
        if times is not None and len(times) and int(times[0]) >= 0:
            bound = int(times[-1])
        # both promises hold for batches that ARE contiguous slices of this time-sorted store; _call_step makes them per call
        self._store_promise = (bound, 1 if times is not None else 0)
        self._step.ts_bound, self._step.sorted_ts = self._store_promise

    def fuses_first_hops(self) -> bool:
        """Did the last call run hop 0 and hop 1 as one launch?  (``tgmx_recency_step_plan`` on the argument block of
        that call; informational -- ``bench.py`` attributes the timed launch's bytes with it.)"""
        if self._step is None:
            return False
        return bool(_native.load().tgmx_recency_step_plan(self._step) & 1)

    def check(self) -> None:
        """Raise the ``ValueError`` the reference would have raised for bad seeds
        seen since the last check (one device->host read)."""
        if self._status is None:
            return
        st = int(self._status.item())
        if st:
            self._status.zero_()
            self._raise_status(st)

    def _raise_status(self, st: int) -> None:
        if st & _ST_SEED_RANGE:
            raise ValueError(f'Seed nodes must satisfy 0 <= x < {self._num_nodes}')
        if st & _ST_SEED_TIME:
            raise ValueError('Seed times must be >= 0')
        if st & _ST_EDGE_RANGE:
            raise ValueError(f'Batch edge endpoints must satisfy 0 <= x < {self._num_nodes}')
        if st & _ST_TS_BOUND:
            raise RuntimeError('tgmx_recency_step: a batch timestamp lies outside [0, ts_bound], or the batch is not time-sorted (both promises come from the graph store)')
        if st & _ST_SCRATCH:
            raise RuntimeError('tgmx_recency_step: the update scratch was not zero-initialised (internal error)')

    # ------------------------------------------------------------------
    def _warn_rings_under_world(self) -> None:
        """One process per GPU with streaming rings: every rank has to replay the WHOLE global batch's update on its ring replica
        (the serial fraction: at 8 ranks and 4096 edges per rank that is a 65 536-entry update per step, modelled 4.6-4.8 x instead
        of 7.6 x at 8 GPUs).  The static index has no state to replicate: say so once."""
        try:
            import torch.distributed as dist

            world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        except Exception:
            world = 1
        if world > 1 and not getattr(RecencyNeighborHook, '_warned_world', False):
            RecencyNeighborHook._warned_world = True
            warnings.warn(
                f"RecencyNeighborHook(mode='ring') under torch.distributed (world size {world}): every rank replays the whole global batch's "
                "ring update, which does not shrink with the number of GPUs.  For multi-GPU jobs build the static index instead -- "
                "RecencyNeighborHook(..., mode='csr', batch_size=<loader batch size>): same sampled neighbours, no per-batch state, and whole "
                'batches can be dealt to ranks with DGDataLoader(batch_shard=(rank, world)).  (INTEGRATION.md section 4)',
                UserWarning, stacklevel=3)

    def _ensure_state(self, dg: DGraph, device: torch.device) -> None:
        if self._edge_x_dim is None:
            self._edge_x_dim = dg.edge_x_dim or 0
        if self._device == device and self._status is not None:
            return
        if device.type != 'cuda':
            raise _native.NativeLibraryError(
                f'RecencyNeighborHook got a batch on {device}; tgm_amd kernels run only on a ROCm device '
                "(no CPU fallback). Create the DGraph with device='cuda'."
            )
        _native.load()
        N, B, D = self._num_nodes, self._max_nbrs, self._edge_x_dim
        old = self._device
        self._device = device
        if self._status is None or old != device:
            self._status = torch.zeros(1, dtype=torch.int32, device=device)
        if self._mode == 'ring':
            if self._ring is None:
                self._warn_rings_under_world()
                self._ring = torch.empty((N * B, 2), dtype=torch.int64, device=device)
                self._ring_x = torch.empty((N * B, max(D, 1)), dtype=torch.float32, device=device) if D else None
                self._write_pos = torch.empty(N, dtype=torch.int32, device=device)
                self.reset_state()
            elif old != device:  # migrate state (recency.py:401-408)
                self._ring = self._ring.to(device)
                self._ring_x = None if self._ring_x is None else self._ring_x.to(device)
                self._write_pos = self._write_pos.to(device)
                self._scratch = None
                self._scratch_edges = 0
        st = _native.RecencyStep()
        if self._mode == 'ring':
            st.ring, st.write_pos, st.ring_x = self._ring.data_ptr(), self._write_pos.data_ptr(), _native.ptr(self._ring_x)
        else:
            self._csr = None  # rebuilt (and its pointers re-bound) on the new device
        st.D, st.B, st.num_nodes = D, B, N
        for hop, k in enumerate(self._num_nbrs[: _native.MAX_HOPS]):
            st.k[hop] = k
        st.directed, st.key_wrap32 = (1 if self._directed else 0), self._key_wrap32
        st.status = self._status.data_ptr()
        st.timed_hop = -1
        self._new_empty = tuple(torch.empty(0, dtype=t, device=device).new_empty for t in (torch.int32, torch.int64, torch.float32))
        self._step = st
        self._bound_store = None
        self._refresh_ts_bound(dg)

    def _ensure_csr(self, dg: DGraph, first_edge: int) -> TemporalCSR:
        store = dg._storage
        if self._csr is None or self._csr_store is not store or self._csr.device != self._device:
            arr = store.on(self._device)
            # the index merges everything before ``first`` into one leading batch (csr.hip): ``first`` must be the SCHEDULE's first
            # edge -- the epoch anchor a batch-sharded loader hands over -- not the first batch this rank happens to take
            first = self._epoch_lo if self._epoch_lo is not None else (first_edge or 0)
            self._csr_first = first
            self._csr_bounds = None if self._batch_starts is None else frozenset(int(b) for b in self._batch_starts)
            self._csr = build_csr(
                arr.src,
                arr.dst,
                arr.ts,
                self._num_nodes,
                batch_starts=self._batch_starts,
                batch_size=self._batch_size,
                first_edge=first,
                directed=self._directed,
            )
            self._csr_store = store
            st = self._step
            st.indptr, st.ring, st.ring_x = self._csr.indptr.data_ptr(), self._csr.adj.data_ptr(), _native.ptr(arr.edge_x)
            # feature rows in adjacency order (one copy of edge_x[adj.eid] at build time: 2E x D floats, 5.6 GB of 288 for the
            # comment-shaped stream): a node's window then gathers consecutive rows instead of rows scattered by edge id
            self._csr_adj_x = None
            st.csr_x_by_pos = 0
            # Not with edge_features='by_id' (the lookups copy no feature row, so the second copy would never be read), not above the
            # explicit cap, and not when it would take more than half of what the device has free right now (the copy itself + the
            # int64 index of the gather); if the allocation fails anyway, the lookups gather by edge id (csr_x_by_pos = 0).
            want = arr.edge_x is not None and self._adj_features and not self._by_id
            if want:
                nbytes = self._csr.adj.shape[0] * (arr.edge_x.shape[1] * 4 + 8)
                free = torch.cuda.mem_get_info(self._device)[0]
                want = nbytes <= self._adj_features_max_bytes and nbytes <= free // 2
            if want:
                try:
                    eid = self._csr.records()[1].to(torch.int64)
                    self._csr_adj_x = arr.edge_x.index_select(0, eid)
                    del eid
                    st.ring_x, st.csr_x_by_pos = self._csr_adj_x.data_ptr(), 1
                except torch.OutOfMemoryError:
                    self._csr_adj_x = None
                    st.ring_x, st.csr_x_by_pos = _native.ptr(arr.edge_x), 0
            # per node: where its visible prefix ended at its last lookup -- a search hint the kernels keep (tgmx_recency_step_t.csr_cursor)
            self._csr_cursor = torch.zeros(self._num_nodes, dtype=torch.int64, device=self._device)
            st.csr_cursor = self._csr_cursor.data_ptr()
        return self._csr

    def _check_csr_boundary(self, ev_hi: int) -> None:
        """The static index orders a node's entries by (batch, time, role, eid): the entries before edge ``ev_hi`` form
        a prefix -- which is what the lookup kernel assumes -- only when ``ev_hi`` is a boundary of the schedule the index
        was built for.  Anything else (another batch size, a split that starts inside a batch, time-unit
        batching with a size-based index) would silently return wrong neighbours: refuse it."""
        if self._csr_bounds is not None:
            ok = ev_hi in self._csr_bounds or ev_hi >= self._csr.num_edges
        else:
            ok = ev_hi >= self._csr_first and (ev_hi - self._csr_first) % self._batch_size == 0 or ev_hi >= self._csr.num_edges
        if not ok:
            raise ValueError(
                f"mode='csr': the batch starts at edge {ev_hi}, which is not a boundary of the schedule the index was built for "
                f'(first edge {self._csr_first}, ' + (f'batch_size {self._batch_size}' if self._csr_bounds is None else 'explicit batch_starts')
                + "). Build the hook with this loader's batch_size / batch_starts, or use mode='ring'."
            )

    def _ensure_scratch(self, n_edges: int, device: torch.device) -> None:
        """Ring-update scratch for batches of up to ``n_edges`` edges (grown, never shrunk)."""
        if n_edges > self._scratch_edges:
            need = int(_native.load().tgmx_ring_update_scratch_bytes(n_edges, 1 if self._directed else 0))
            if need == 0:
                _native.check(-2, 'tgmx_ring_update_scratch_bytes')
            # torch allocations are 256-byte aligned; zeros: the head of the scratch holds a self-resetting barrier
            self._scratch = torch.zeros(need, dtype=torch.uint8, device=device)
            self._scratch_edges = n_edges
            self._step.scratch = self._scratch.data_ptr()

    def _is_store_slice(self, batch: DGBatch, tt: Tensor, device: torch.device) -> bool:
        lo, store = batch._edge_lo, self._bound_store
        if lo is None or store is None or not hasattr(store, 'on'):
            return False
        base = store.on(device).ts
        return tt.data_ptr() == base.data_ptr() + 8 * int(lo) and lo + tt.shape[0] <= base.shape[0]

    def _note_batch_time(self, dg: DGraph, batch: DGBatch) -> None:
        """The reference warns when a query is older than everything the buffers hold (recency.py:242-251:
        ``query_times.min() < self._nbr_times.min()``, a scan of the whole [N, B] buffer per hop -- whose minimum is 0 until
        every slot of every node has been written, so it practically never fires).  What it guards against is batches
        arriving out of chronological order; that is tracked here in O(1) from the store's host timeline: warn (once per
        epoch) when a batch starts before the previous one did."""
        lo = batch._event_lo
        store = getattr(dg, '_storage', None)
        times = getattr(store, '_time_np', None)
        if lo is None or times is None or lo >= len(times):
            return
        t0 = int(times[lo])
        if t0 < self._last_batch_t and not self._warned_order:
            self._warned_order = True
            logger.warning(
                f'RecencyNeighborHook: this batch starts at t={t0}, behind the previous batch (t={self._last_batch_t}). '
                'Results may be incomplete or incorrect. This hook assumes queries are processed in chronological order: '
                '(1) the dataloader must return batches sorted by timestamp, (2) batches must not be shuffled, '
                '(3) reset the hook state between datasets / epochs / evaluation runs.'
            )
        self._last_batch_t = t0

    # ------------------------------------------------------------------
    def __call__(self, dg: DGraph, batch: DGBatch) -> DGBatch:
        return self._call_step(dg, batch, batch.edge_src.device)

    # ------------------------------------------------------------------
    def _publish(self, batch: DGBatch, seed_n, seed_t, out_n, out_t, out_x, seed_mask) -> DGBatch:
        # the lists of one call carry its tag (plain lists otherwise): tgm_amd.nn.TGAT recognises hops sampled for one another by it
        from ..core.lazy import EdgeFeaturesById, SampledHops, SamplerCallTag

        tag = SamplerCallTag(out_n, out_t, out_x)
        if isinstance(out_x, EdgeFeaturesById):
            out_x.tag = tag
        else:
            out_x = SampledHops(out_x, tag)
        self.add_batch_attribute(batch, 'seed_nids', SampledHops(seed_n, tag))
        self.add_batch_attribute(batch, 'seed_times', SampledHops(seed_t, tag))
        self.add_batch_attribute(batch, 'nbr_nids', SampledHops(out_n, tag))
        self.add_batch_attribute(batch, 'nbr_edge_time', SampledHops(out_t, tag))
        self.add_batch_attribute(batch, 'nbr_edge_x', out_x)
        self.add_batch_attribute(batch, 'seed_node_nbr_mask', seed_mask)
        return batch

    def _call_step(self, dg: DGraph, batch: DGBatch, device: torch.device) -> DGBatch:
        """The whole call (seed concat, every hop, ring update in streaming mode) is ONE tgmx_recency_step."""
        groups, group_times, seed_mask = self._get_seed_tensors(batch, device, concat=False)
        S0 = 0
        for g in groups:
            S0 += g.shape[0]
        L = len(self._num_nbrs)
        if S0 == 0:
            # reference: CPU empties, and the update is skipped (recency.py:127-139)
            D0 = dg.edge_x_dim or 0
            return self._publish(
                batch,
                [torch.empty(0, dtype=torch.int32) for _ in range(L)],
                [torch.empty(0, dtype=torch.int64) for _ in range(L)],
                [torch.empty(0, dtype=torch.int32) for _ in range(L)],
                [torch.empty(0, dtype=torch.int64) for _ in range(L)],
                [torch.empty(0, D0, dtype=torch.float32) for _ in range(L)],
                seed_mask,
            )
        if len(groups) > _native.MAX_SEED_GROUPS or L > _native.MAX_HOPS:
            raise ValueError(f'at most {_native.MAX_SEED_GROUPS} seed groups and {_native.MAX_HOPS} hops are supported')
        self._ensure_state(dg, device)
        self._note_batch_time(dg, batch)
        if getattr(dg, '_storage', None) is not self._bound_store:  # another graph: its timestamps may be larger
            self._refresh_ts_bound(dg)
        D = self._edge_x_dim
        st = self._step
        lib = _native.load()
        empty = torch.empty
        ring_mode = self._mode == 'ring'
        with _on_device(device):
            stream = _native.stream_ptr(device.index)
            if not ring_mode:
                if batch._edge_lo is None:
                    raise ValueError("mode='csr' needs batches materialized by tgm_amd.DGraph (batch._edge_lo is unset)")
                ev_hi = int(batch._edge_lo)
                self._ensure_csr(dg, ev_hi)
                self._check_csr_boundary(ev_hi)
                if self._epoch_lo is None:
                    self._epoch_lo = ev_hi
                st.ev_lo, st.ev_hi = self._epoch_lo, ev_hi
            # hop-0 seeds (fresh tensors owned by the batch, like the reference's torch.cat)
            # bound new_empty of per-dtype prototypes: ~0.5 us less per tensor than torch.empty(..., dtype=, device=)
            ne32, ne64, nef = self._new_empty
            seeds = ne32(S0)
            seed_times = ne64(S0)
            for g, (a, t) in enumerate(zip(groups, group_times)):
                if not a.is_contiguous():
                    a = a.contiguous()
                if not t.is_contiguous():
                    t = t.contiguous()
                if t.shape[0] != a.shape[0]:
                    raise ValueError(f'seed group {g}: {a.shape[0]} nodes but {t.shape[0]} times')
                st.grp_nid[g], st.grp_ts[g], st.grp_n[g] = a.data_ptr(), t.data_ptr(), a.shape[0]
            st.n_groups = len(groups)
            st.seed_nid0, st.seed_ts0 = seeds.data_ptr(), seed_times.data_ptr()

            out_seed_n, out_seed_t, out_n, out_t, out_x = [], [], [], [], []
            cur_n, cur_t, S = seeds, seed_times, S0
            by_id = self._by_id and D > 0
            if by_id and ring_mode and batch._edge_lo is None:
                raise ValueError("edge_features='by_id' needs batches materialized from the graph store (their edges carry store ids)")
            out_eid = []
            for hop, k in enumerate(self._num_nbrs):
                nid = ne32((S, k))
                nts = ne64((S, k))
                if by_id:
                    nx = None
                    out_eid.append(ne32((S, k)))
                    st.out_eid[hop] = out_eid[-1].data_ptr()
                else:
                    nx = nef((S, k, D))
                    st.out_eid[hop] = 0
                st.out_nid[hop], st.out_ts[hop], st.out_x[hop] = nid.data_ptr(), nts.data_ptr(), _native.ptr(nx)
                out_seed_n.append(cur_n)
                out_seed_t.append(cur_t)
                out_n.append(nid)
                out_t.append(nts)
                out_x.append(nx)
                cur_n, cur_t = nid.view(-1), nts.view(-1)
                S *= k

            self._calls += 1
            timer = None
            st.timed_hop = -1
            if self.profile_hop is not None and self._calls % self.profile_every == 0 and self.profile_pool:
                timer = self.profile_pool.pop()
                st.timed_hop, st.ev_start, st.ev_stop = self.profile_hop, timer.start, timer.stop

            n_edges = batch.edge_src.shape[0] if ring_mode else 0
            keep = None
            if n_edges:
                self._ensure_scratch(n_edges, device)
                ex = batch.edge_x
                if ex is not None and D:
                    if ex.dtype != torch.float32 or not ex.is_contiguous():
                        ex = ex.to(torch.float32).contiguous()
                else:
                    ex = None
                src, dst, tt = batch.edge_src, batch.edge_dst, batch.edge_time
                if not (src.is_contiguous() and dst.is_contiguous() and tt.is_contiguous()):
                    src, dst, tt = src.contiguous(), dst.contiguous(), tt.contiguous()
                keep = (src, dst, tt, ex)
                st.src, st.dst, st.ts, st.edge_x = src.data_ptr(), dst.data_ptr(), tt.data_ptr(), _native.ptr(ex)
                st.eid0 = -1 if batch._edge_lo is None else int(batch._edge_lo)
                # "time-sorted, 0 <= t <= ts_bound" is the STORE's promise: it covers this batch only if the batch is a zero-copy
                # slice of the store (a user-built, filtered or permuted batch gets the span reduction and no sortedness claim)
                st.ts_bound, st.sorted_ts = self._store_promise if self._is_store_slice(batch, tt, device) else (0, 0)

            # bad seeds must leave the state untouched (the reference validates before it changes anything): with
            # guard_seed_errors the update of this very call skips its writes when the lookups flagged a seed, so one call and
            # ONE read-back per batch reproduce "raise, state unchanged".  'sync' only: the status word is sticky until check()
            # zeroes it, so under 'deferred' a guard would silently drop every later batch's update after one bad seed
            st.n, st.n_hops = n_edges, L
            st.guard_seed_errors = 1 if self._validate == 'sync' else 0
            rc = lib.tgmx_recency_step(st, stream)
            if rc:
                _native.check(rc, 'tgmx_recency_step')
            if self._validate == 'sync':
                self.check()
            del keep
            if timer is not None:
                # every hop the timed launch covers: (seed rows, k) and the number of valid neighbor slots per hop as a
                # small device tensor -- holding the outputs themselves would pin allocator blocks and slow later steps
                hops = [0, 1] if self.profile_hop in (0, 1) and self.fuses_first_hops() else [self.profile_hop]
                counts = torch.empty(len(hops), dtype=torch.int64, device=device)
                for i, h in enumerate(hops):
                    torch.sum((out_n[h] != -1).view(-1), dim=0, dtype=torch.int64, out=counts[i])  # asynchronous, unlike count_nonzero
                self.profile_log.append((timer, [(out_seed_n[h].shape[0], self._num_nbrs[h]) for h in hops], counts, None))
        if by_id:
            from ..core.lazy import EdgeFeaturesById

            out_x = EdgeFeaturesById(out_eid, self._feature_table(dg, device))
        return self._publish(batch, out_seed_n, out_seed_t, out_n, out_t, out_x, seed_mask)

    def _feature_table(self, dg: DGraph, device: torch.device) -> Tensor:
        """[E, D] edge features of the resident store: what the edge ids of edge_features='by_id' index."""
        return dg._storage.on(device).edge_x

    # ------------------------------------------------------------------
    def _get_seed_tensors(self, batch: DGBatch, device: torch.device, concat: bool = True):
        """Concatenate the hop-0 seeds (recency.py:173-237).  Structural checks
        (missing / non-tensor / non-1-D) raise here; value checks (id range,
        negative time) are done by the lookup kernel."""
        seeds: List[Tensor] = []
        times: List[Tensor] = []
        mask: Dict[str, Tensor] = {}
        offset = 0
        for node_attr, time_attr in zip(self._seed_nodes_keys, self._seed_times_keys):
            missing = [a for a in (node_attr, time_attr) if not hasattr(batch, a)]
            if missing:
                raise ValueError(f'Missing seed attributes {missing} on batch')
            seed, time = getattr(batch, node_attr), getattr(batch, time_attr)
            skip = False
            for name, tensor in ((node_attr, seed), (time_attr, time)):
                if tensor is None:
                    if not self._warned_seed_None:
                        warnings.warn(
                            f'Seed attribute {name} is None on this batch, skipping this batch. '
                            'Future occurrences will also be skipped but the warning will be suppressed',
                            UserWarning,
                        )
                        self._warned_seed_None = True
                    skip = True
                    break
                if not isinstance(tensor, Tensor):
                    raise ValueError(f'{name} must be a Tensor, got {type(tensor)}')
                if tensor.ndim != 1:
                    raise ValueError(f'{name} must be 1-D, got shape {tensor.shape}')
            if skip:
                continue
            if device.type != 'cuda' and self._validate != 'off' and seed.numel():
                # host tensors: the value checks are free, do them eagerly like the reference
                if bool((seed < 0).any()) or bool((seed >= self._num_nodes).any()):
                    raise ValueError(
                        f'Seed nodes in {node_attr} must satisfy 0 <= x < {self._num_nodes}, '
                        f'got values in range [{seed.min().item()}, {seed.max().item()}]'
                    )
                if bool((time < 0).any()):
                    raise ValueError(f'Seed times in {time_attr} must be >= 0, got min value: {time.min().item()}')
            if seed.device != device or seed.dtype != torch.int32:
                seed = seed.to(device=device, dtype=torch.int32)
            if time.device != device or time.dtype != torch.int64:
                time = time.to(device=device, dtype=torch.int64)
            seeds.append(seed)
            times.append(time)
            n = seed.shape[0]
            mask[node_attr] = (offset, n)
            offset += n
        # fresh per batch, like the reference (recency.py:221-224): ONE arange (one launch) and a window of it per key
        if offset and device.type == 'cuda':
            whole = torch.arange(offset, device=device)
            mask = {k: whole.narrow(0, o, n) for k, (o, n) in mask.items()}
        else:
            mask = {k: torch.arange(o, o + n, device=device) for k, (o, n) in mask.items()}
        if not concat:
            return seeds, times, mask  # ring mode: the groups are concatenated by the step call itself
        if seeds:
            return torch.cat(seeds), torch.cat(times), mask  # fresh tensors owned by the batch
        return (
            torch.empty(0, dtype=torch.int32, device=device),
            torch.empty(0, dtype=torch.int64, device=device),
            mask,
        )

"""The loader's per-batch hook chain lowered to ONE native call with pooled outputs.

At the headline shape a batch is ~40 us of kernels; walking the chain in Python
(slice view -> ``materialize`` -> ``HookManager`` -> one call per hook, a torch
allocation per output) costs more than that on the host, so the GPU idles.
``DGDataLoader(..., output_pool=R)`` asks for the chain's lowerable prefix

    [EdgeShardHook] -> [RandomNegativeEdgeSamplerHook] -> RecencyNeighborHook -> [DeduplicationHook -> [SampledEdgeListHook]]

to run as one ``tgmx_pipeline_step`` (``include/tgm_amd.h``; it restates
tgm/data/loader.py:158-170 + tgm/hooks/hook_manager.py:139-168 for that chain)
writing into a ring of ``R`` preallocated output sets.  Hooks behind the prefix
run as usual.  The hooks stay the owners of their state (rings, static index,
RNG call counter, status word), so ``reset_state`` / ``check`` and switching
between the lowered and the hook-by-hook path keep working, and both paths
produce identical tensors (``tests/test_pipeline_gpu.py``).

Contract of ``output_pool=R`` (the only difference to the reference's
semantics): the tensors a batch carries are recycled -- they stay valid until
``R`` more batches have been produced on the same stream; ``R=0`` (default)
hands out fresh tensors per batch like the reference.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Sequence

import torch

from . import _native
from .core import DGBatch, DGraph
from .dist import EdgeShardHook, shard_bounds
from .hooks.dedup import DeduplicationHook
from .hooks.edge_list import SampledEdgeListHook
from .hooks.negatives import RandomNegativeEdgeSamplerHook
from .hooks.recency import RecencyNeighborHook

_ROLE_KEYS = {
    False: {'edge_src': (_native.SEED_SRC, 'edge_time'), 'edge_dst': (_native.SEED_DST, 'edge_time'), 'neg': (_native.SEED_NEG, 'neg_time')},
    True: {'shard_src': (_native.SEED_SRC, 'shard_time'), 'shard_dst': (_native.SEED_DST, 'shard_time'), 'neg': (_native.SEED_NEG, 'neg_time')},
}


class _Slot:
    """One preallocated output set + its filled ``tgmx_pipeline_out_t`` + the attributes it puts on a batch."""

    __slots__ = ('out', 'attrs', 'tensors', 'nbr_nids', 'post', 'post_bufs', 'valid')


class CompiledPipeline:
    def __init__(self, dg: DGraph, shard: Optional[EdgeShardHook], neg: Optional[RandomNegativeEdgeSamplerHook],
                 nbr: RecencyNeighborHook, n_lowered: int, pool: int, dedup: Optional[DeduplicationHook] = None,
                 edges: Optional[SampledEdgeListHook] = None) -> None:  # fmt: skip
        self.n_lowered = n_lowered  # hooks of the chain this object replaces
        self._dedup, self._edges = dedup, edges
        self._dedup_ws = None
        self._dg, self._shard, self._neg, self._nbr = dg, shard, neg, nbr
        self._R = max(1, int(pool))
        # delta feature writes into the pooled outputs (include/tgm_amd.h: tgmx_recency_step_t.out_valid); TGMX_DELTA_WRITES=0: off
        self._delta = os.environ.get('TGMX_DELTA_WRITES', '1') != '0'
        self._pools: Dict[int, List[_Slot]] = {}
        self._turn = 0
        self._pipe: Optional[_native.Pipeline] = None
        self._step_ref = None  # the hook's argument block this pipeline was bound to
        self._lib = _native.load()
        self._arr = dg._storage.on(dg.device)
        self._device = self._arr.src.device  # with its index ('cuda' -> 'cuda:0'): what the hooks see on batch tensors
        self._roles = [_ROLE_KEYS[shard is not None][k][0] for k in nbr._seed_nodes_keys]

    # -- lowering ---------------------------------------------------------------
    @staticmethod
    def lower(dg: DGraph, hooks: Sequence, pool: int) -> Optional['CompiledPipeline']:
        """The pipeline for the longest lowerable prefix of ``hooks`` (None: nothing to lower)."""
        if pool <= 0 or dg.device.type != 'cuda' or not hasattr(dg, '_storage'):
            return None
        i = 0
        shard = neg = None
        if i < len(hooks) and type(hooks[i]) is EdgeShardHook and hooks[i]._id is None:
            shard = hooks[i]
            i += 1
        if i < len(hooks) and type(hooks[i]) is RandomNegativeEdgeSamplerHook:
            h = hooks[i]
            want = ('shard_dst', 'shard_time') if shard is not None else ('edge_dst', 'edge_time')
            if h._id is None and h.neg_ratio == 1.0 and (h._like, h._time_key) == want:
                neg = h
                i += 1
        if i >= len(hooks) or type(hooks[i]) is not RecencyNeighborHook:
            return None
        nbr = hooks[i]
        keys = _ROLE_KEYS[shard is not None]
        ok = (
            nbr._id is None
            and len(nbr._num_nbrs) <= _native.MAX_HOPS
            and 1 <= len(nbr._seed_nodes_keys) <= _native.MAX_SEED_GROUPS
            and all(k in keys and keys[k][1] == t for k, t in zip(nbr._seed_nodes_keys, nbr._seed_times_keys))
            and (neg is not None) == ('neg' in nbr._seed_nodes_keys)  # negatives only exist as a seed role of the lowered step
            and nbr._seed_nodes_keys.count('neg') <= 1
        )
        if not ok:
            return None
        i += 1
        # the TGN tail: unique ids over [src | dst | neg | sampled neighbors], then the compact edge list of one hop
        dedup = edges = None
        if i < len(hooks) and type(hooks[i]) is DeduplicationHook and hooks[i]._id is None:
            extra = set(hooks[i].seed_keys or [])
            if extra <= {'neg', 'nbr_nids'} and ('neg' not in extra or neg is not None) and len(nbr._num_nbrs) + 3 <= 16:
                dedup = hooks[i]
                i += 1
                if i < len(hooks) and type(hooks[i]) is SampledEdgeListHook and hooks[i]._id is None and hooks[i].hop < len(nbr._num_nbrs):
                    edges = hooks[i]
                    i += 1
        return CompiledPipeline(dg, shard, neg, nbr, i, pool, dedup, edges)

    def _bind(self) -> None:
        """(Re)build the native argument block from the hooks' current state."""
        nbr, arr = self._nbr, self._arr
        nbr._ensure_state(self._dg, self._device)
        p = _native.Pipeline()
        p.src, p.dst, p.ts, p.edge_x = arr.src.data_ptr(), arr.dst.data_ptr(), arr.ts.data_ptr(), _native.ptr(arr.edge_x)
        p.num_edges = arr.src.shape[0]
        p.rank, p.world = (self._shard.rank, self._shard.world_size) if self._shard is not None else (0, 1)
        p.n_roles = len(self._roles)
        for g, r in enumerate(self._roles):
            p.seed_role[g] = r
        if self._neg is not None:
            neg = self._neg
            if neg._rng_seed is None:
                neg._rng_seed = (neg._seed if neg._seed is not None else torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
            p.neg_low, p.neg_high, p.neg_seed = neg.low, neg.high, neg._rng_seed
        p.update = 1 if nbr._mode == 'ring' else 0
        ctypes.memmove(ctypes.byref(p.step), ctypes.byref(nbr._step), ctypes.sizeof(_native.RecencyStep))
        p.step.n_hops = len(nbr._num_nbrs)
        p.step.guard_seed_errors = 1 if nbr._validate != 'off' else 0
        self._pipe, self._step_ref = p, nbr._step
        self._scratch_ptr = nbr._step.scratch

    def _make_pool(self, n: int) -> List[_Slot]:
        nbr, dev = self._nbr, self._device
        lo, hi = shard_bounds(n, *((self._shard.rank, self._shard.world_size) if self._shard is not None else (0, 1)))
        share = hi - lo
        D = nbr._edge_x_dim
        S0 = share * len(self._roles)
        offsets = {k: g * share for g, k in enumerate(nbr._seed_nodes_keys)}
        slots = []
        for _ in range(self._R):
            sl = _Slot()
            out = _native.PipelineOut()
            out.timed_hop = -1
            attrs: dict = {}
            tensors = []
            if self._neg is not None:
                neg_t = torch.empty(share, dtype=torch.int32, device=dev)
                negt_t = torch.empty(share, dtype=torch.int64, device=dev)
                out.neg, out.neg_time = neg_t.data_ptr(), negt_t.data_ptr()
                attrs['neg'], attrs['neg_time'] = neg_t, negt_t
            seeds = torch.empty(S0, dtype=torch.int32, device=dev)
            seed_t = torch.empty(S0, dtype=torch.int64, device=dev)
            out.seed_nid0, out.seed_ts0 = seeds.data_ptr(), seed_t.data_ptr()
            seed_n, seed_ts, nbr_n, nbr_t, nbr_x = [], [], [], [], []
            cur_n, cur_t, S = seeds, seed_t, S0
            valid = []
            for hop, k in enumerate(nbr._num_nbrs):
                nid = torch.empty((S, k), dtype=torch.int32, device=dev)
                nts = torch.empty((S, k), dtype=torch.int64, device=dev)
                if self._delta:
                    # persistent buffers: the lookups write a feature row only from the first slot that changes (its valid slots
                    # are the right-aligned tail, the rest is zero and stays zero) -- tgmx_recency_step_t.out_valid.  The
                    # buffers start as all-pad rows; consumers must treat them as read-only.
                    nx = torch.zeros((S, k, D), dtype=torch.float32, device=dev)
                    nv = torch.zeros((2, S), dtype=torch.int32, device=dev)  # [0]: the rows' spans, [1]: the spans before the last call
                    out.out_valid[hop], out.out_valid_prev[hop] = nv[0].data_ptr(), nv[1].data_ptr()
                    valid.append(nv)
                else:
                    nx = torch.empty((S, k, D), dtype=torch.float32, device=dev)
                out.out_nid[hop], out.out_ts[hop], out.out_x[hop] = nid.data_ptr(), nts.data_ptr(), nx.data_ptr()
                seed_n.append(cur_n)
                seed_ts.append(cur_t)
                nbr_n.append(nid)
                nbr_t.append(nts)
                nbr_x.append(nx)
                cur_n, cur_t = nid.view(-1), nts.view(-1)
                S *= k
            whole = torch.arange(S0, device=dev)
            attrs.update(seed_nids=seed_n, seed_times=seed_ts, nbr_nids=nbr_n, nbr_edge_time=nbr_t, nbr_edge_x=nbr_x,
                         seed_node_nbr_mask={k: whole.narrow(0, o, share) for k, o in offsets.items()})  # fmt: skip
            sl.out, sl.attrs, sl.nbr_nids, sl.valid = out, attrs, nbr_n, valid
            sl.post = sl.post_bufs = None
            if self._dedup is not None:
                sl.post, sl.post_bufs = self._make_post(n, share, nbr_n, dev)
            slots.append(sl)
        self._pools[n] = slots
        return slots

    def _make_post(self, n: int, share: int, nbr_n, dev):
        """tgmx_pipeline_post_t of one slot: unique-id and edge-list buffers, the size mirror in pinned memory, its event."""
        nbr = self._nbr
        N = int(self._dg._storage.num_nodes_global)
        extra = set(self._dedup.seed_keys or [])
        total = 2 * n + (share if 'neg' in extra else 0) + (sum(t.numel() for t in nbr_n) if 'nbr_nids' in extra else 0)
        if self._dedup_ws is None:
            need = int(self._lib.tgmx_unique_ids_workspace_bytes(N))
            self._dedup_ws = torch.zeros(need, dtype=torch.uint8, device=dev)  # zeros: the bitmap cleans itself
        uniq = torch.empty(max(min(total, N), 1), dtype=torch.int32, device=dev)
        dev_sizes = torch.zeros(3, dtype=torch.int64, device=dev)
        pin = torch.zeros(3, dtype=torch.int64).pin_memory()
        ev = ctypes.c_void_p()
        _native.check(self._lib.tgmx_event_create(ctypes.byref(ev)), 'tgmx_event_create')
        post = _native.PipelinePost()
        post.dedup, post.dedup_neg, post.dedup_nbr, post.num_nodes = 1, int('neg' in extra), int('nbr_nids' in extra), N
        post.dedup_ws, post.uniq_out = self._dedup_ws.data_ptr(), uniq.data_ptr()
        post.edge_hop = -1
        ei = et = ex = ro = None
        if self._edges is not None:
            h = self._edges.hop
            S, k = nbr_n[h].shape
            D = nbr._edge_x_dim
            cap = max(S * k, 1)
            ei = torch.empty((2, cap), dtype=torch.int64, device=dev)
            et = torch.empty(cap, dtype=torch.int64, device=dev)
            ex = torch.empty((cap, D), dtype=torch.float32, device=dev)
            ro = torch.empty(S + 1, dtype=torch.int64, device=dev)
            post.edge_hop, post.edge_cap = h, cap
            post.row_off, post.edge_index, post.edge_t, post.edge_x = ro.data_ptr(), ei.data_ptr(), et.data_ptr(), ex.data_ptr()
        post.dev_sizes, post.host_sizes, post.sizes_ready = dev_sizes.data_ptr(), pin.data_ptr(), ev.value
        return post, (uniq, dev_sizes, pin, ev.value, ei, et, ex, ro, N)

    def _defer_post(self, batch: DGBatch, slot: _Slot) -> None:
        uniq, dev_sizes, pin, ev, ei, et, ex, _, N = slot.post_bufs
        dedup, edges, lib = self._dedup, self._edges, self._lib
        batch.__dict__['_unique_dev'] = (uniq, dev_sizes[0:1])

        def finish() -> None:
            _native.check(lib.tgmx_event_synchronize(ev), 'tgmx_event_synchronize')  # the only wait: three sizes
            cnt, st, E = pin.tolist()
            if st & 0xFFFFFFFF:
                dev_sizes[1].zero_()
                raise ValueError(f'node ids must satisfy 0 <= x < {N} (or -1 for a padded neighbor slot)')
            dedup._publish(batch, uniq[:cnt])
            if edges is not None:
                edges.add_batch_attribute(batch, 'sampled_edge_index', ei[:, :E])
                edges.add_batch_attribute(batch, 'sampled_edge_time', et[:E])
                edges.add_batch_attribute(batch, 'sampled_edge_x', ex[:E])

        batch._defer(finish)

    # -- per batch ----------------------------------------------------------------
    def step(self, lo: int, n: int, batch: DGBatch) -> bool:
        """Run the lowered prefix for the batch = edges [lo, lo + n) and put its outputs on ``batch``.
        False: this batch is not handled here (empty share: the hooks' own empty-batch behaviour applies)."""
        nbr = self._nbr
        if nbr._step is not self._step_ref or self._pipe is None:
            self._bind()
        shard = self._shard
        if shard is not None:
            s_lo, s_hi = shard_bounds(n, shard.rank, shard.world_size)
            if s_hi == s_lo:
                return False
        elif n == 0:
            return False
        slots = self._pools.get(n)
        if slots is None:
            if len(self._pools) >= 8:  # time-unit batching: every batch may have its own size -- keep the 8 most recent shapes
                self._pools.pop(next(iter(self._pools)))
            slots = self._make_pool(n)
        turn = self._turn
        self._turn = turn + 1
        slot = slots[turn % self._R]
        pipe = self._pipe
        ring_mode = nbr._mode == 'ring'
        if ring_mode:
            if n > nbr._scratch_edges:
                nbr._ensure_scratch(n, self._device)
            if nbr._step.scratch != self._scratch_ptr:
                self._scratch_ptr = pipe.step.scratch = nbr._step.scratch
        else:
            nbr._ensure_csr(self._dg, lo)
            nbr._check_csr_boundary(lo)
            if nbr._epoch_lo is None:
                nbr._epoch_lo = lo
            st = pipe.step
            st.indptr, st.ring, st.ring_x, st.ev_lo = nbr._step.indptr, nbr._step.ring, nbr._step.ring_x, nbr._epoch_lo
        nbr._note_batch_time(self._dg, batch)
        call = 0
        neg = self._neg
        if neg is not None:
            call = neg._calls = neg._calls + 1
        out = slot.out
        nbr._calls += 1
        timer = None
        if nbr.profile_hop is not None and nbr._calls % nbr.profile_every == 0 and nbr.profile_pool:
            timer = nbr.profile_pool.pop()
            out.timed_hop, out.ev_start, out.ev_stop = nbr.profile_hop, timer.start, timer.stop
        rc = self._lib.tgmx_pipeline_step(pipe, lo, n, call, out, slot.post, _native.stream_ptr(self._device.index))
        if rc:
            _native.check(rc, 'tgmx_pipeline_step')
        if timer is not None:
            out.timed_hop = -1
            self._log_timed(timer, slot)
        if nbr._validate == 'sync':
            nbr.check()  # one device -> host read per batch: the reference's raise-per-call behaviour
        d = batch.__dict__
        if shard is not None:
            arr = self._arr
            d['shard_src'] = arr.src.narrow(0, lo + s_lo, s_hi - s_lo)
            d['shard_dst'] = arr.dst.narrow(0, lo + s_lo, s_hi - s_lo)
            d['shard_time'] = arr.ts.narrow(0, lo + s_lo, s_hi - s_lo)
            d['shard_lo'] = s_lo
        d.update(slot.attrs)
        if slot.post is not None:
            self._defer_post(batch, slot)
        return True

    def _log_timed(self, timer, slot: _Slot) -> None:
        """What bench.py's byte model needs of a timed launch, as ONE small launch per hop behind it (tgmx_lookup_accounting: partial sums
        the reader adds up after the run): valid slots, and -- delta feature writes -- the slots whose feature row was rewritten."""
        nbr = self._nbr
        st = self._pipe.step
        # what tgmx_recency_step_plan needs of the per-call fields: the seed count and the hop-1 output alignment
        st.n_groups, st.S0 = 0, slot.attrs['seed_nids'][0].shape[0]
        for h in range(len(nbr._num_nbrs)):
            st.out_x[h] = slot.out.out_x[h]
        fused = bool(self._lib.tgmx_recency_step_plan(st) & 1)
        hops = [0, 1] if nbr.profile_hop in (0, 1) and fused else [nbr.profile_hop]
        parts = torch.empty((len(hops), _native.ACCOUNTING_PARTIALS, 3), dtype=torch.int64, device=self._device)
        stream = _native.stream_ptr(self._device.index)
        for i, h in enumerate(hops):
            ids = slot.nbr_nids[h]
            sp = slot.valid[h] if slot.valid else None
            _native.check(self._lib.tgmx_lookup_accounting(ids.data_ptr(), ids.numel(), sp[1].data_ptr() if sp is not None else None,
                                                           sp[0].data_ptr() if sp is not None else None, ids.shape[0], parts[i].data_ptr(), stream),
                          'tgmx_lookup_accounting')
        nbr.profile_log.append((timer, [(slot.attrs['seed_nids'][h].shape[0], nbr._num_nbrs[h]) for h in hops], _Counts(parts, 0),
                                _Counts(parts, 1) if slot.valid else None))


class _Counts:
    """column `col` of tgmx_lookup_accounting's partial sums, one entry per hop -- summed when somebody asks (after the timed region)"""

    def __init__(self, parts: torch.Tensor, col: int) -> None:
        self._parts, self._col = parts, col

    def sum(self) -> torch.Tensor:
        return self._parts[:, :, self._col].sum()

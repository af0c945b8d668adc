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

Output sets.  Every tensor a batch carries is carved out of ONE persistent byte
buffer per output set (one storage), sized for the largest batch seen.

``output_pool=None`` (the loader's default): FRESH-TENSOR SEMANTICS from a pool.
A set is handed out again only when nothing outside the pipeline can reach any
of its tensors any more -- Python reference counts of the handed-out tensors,
their ``TensorImpl`` use counts (views, tensors saved for backward) and the
storage's use count (``detach()``, any other alias) are all back at their
baseline -- otherwise another set is taken (at most ``_MAX_AUTO_SETS`` are kept;
beyond that a batch gets a set nobody keeps: plain fresh tensors).  A loop that
drops a batch before asking for the next one runs on one set, ``for batch in
loader`` alternates between two.  In-place modification through torch ops is
seen in the buffer's version counter and the set is re-initialised before its
next use, so the delta feature writes (below) stay exact.  THE STREAM CONTRACT of this
mode: "no references left" proves that no host code can reach the tensors, not that
device work reading them has finished -- that is guaranteed only for work enqueued on
the stream the loader runs on (the next batch is produced behind it, in stream order).
A consumer that reads a batch on a SIDE stream (or hands it to a collective that relies
on ``record_stream``) must keep a reference to the batch until that work has been
ordered before the loader's stream (``loader_stream.wait_stream(side)``), or ask for
``output_pool=0``.  A set filled on one stream is never recycled from another one.

``output_pool=R`` (explicit): a ring of ``R`` sets recycled unconditionally --
the tensors stay valid until ``R`` more batches have been produced on the same
stream and must be treated as read-only.  ``output_pool=0``: no lowering, the
hooks run one by one and allocate fresh tensors per batch.
"""
from __future__ import annotations

import ctypes
import os
import sys
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


_STATUS_READ_EVERY = 512  # batches between reads of the device status word when the store vouches for the seeds (validate='sync')
_TAGGED = frozenset(('seed_nids', 'seed_times', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x'))
_MAX_AUTO_SETS = 4  # output sets kept by the liveness-checked pool (output_pool=None); a batch beyond that gets an unpooled set
_storage_uses = torch._C._storage_Use_Count


def _snapshot(tensors: list) -> list:
    """(Python references, TensorImpl references) of every handed-out tensor of a set -- ONE code path for the baseline and for
    the check at reuse, so the references the measurement itself holds cancel."""
    return [(sys.getrefcount(t), t._use_count()) for t in tensors]


class _Slot:
    """The tensors one batch of `n` edges carries, as views into an output set's buffer, + the filled ``tgmx_pipeline_out_t``."""

    __slots__ = ('out', 'attrs', 'nbr_nids', 'post', 'post_bufs', 'valid', 'watch', 'base')


class _OutputSet:
    """One persistent output set: a single byte buffer (one storage) that every tensor of a batch is a view of."""

    __slots__ = ('buf', 'cap', 'off', 'views', 'storage', 'uses', 'version', 'pin', 'event', 'arange_n', 'lib', 'stream')

    def __del__(self) -> None:
        ev, lib = getattr(self, 'event', None), getattr(self, 'lib', None)
        if ev and lib is not None:
            try:
                lib.tgmx_event_destroy(ev)
            except Exception:
                pass


class CompiledPipeline:
    def __init__(self, dg: DGraph, shard: Optional[EdgeShardHook], neg: Optional[RandomNegativeEdgeSamplerHook],
                 nbr: RecencyNeighborHook, n_lowered: int, pool: Optional[int], dedup: Optional[DeduplicationHook] = None,
                 edges: Optional[SampledEdgeListHook] = None) -> None:  # fmt: skip
        self.n_lowered = n_lowered  # hooks of the chain this object replaces
        self._dedup, self._edges = dedup, edges
        self._dedup_ws = None
        self._dg, self._shard, self._neg, self._nbr = dg, shard, neg, nbr
        # pool=None: liveness-checked sets (fresh-tensor semantics); pool=R: a ring of R sets recycled unconditionally
        self._safe = pool is None
        self._R = 1 if pool is None else max(1, int(pool))
        # delta feature writes into the persistent outputs (include/tgm_amd.h: tgmx_recency_step_t.out_valid); TGMX_DELTA_WRITES=0: off
        self._delta = os.environ.get('TGMX_DELTA_WRITES', '1') != '0'
        self._sets: List[_OutputSet] = []
        self._cap = 0
        self._last = 0
        self._turn = 0
        self._pipe: Optional[_native.Pipeline] = None
        self._step_ref = None  # the hook's argument block this pipeline was bound to
        self._lib = _native.load()
        self._arr = dg._storage.on(dg.device)
        self._device = self._arr.src.device  # with its index ('cuda' -> 'cuda:0'): what the hooks see on batch tensors
        self._roles = [_ROLE_KEYS[shard is not None][k][0] for k in nbr._seed_nodes_keys]
        self._static_ok: Optional[tuple] = None  # host-side seed validation of the resident store (validate='sync')
        self._since_check = 0
        self._async = None  # (worker, wait event or None, record event): DGDataLoader(side_stream=True) sets it around a call
        self._last_ticket = 0  # the launch worker's newest job of this pipeline (0: none yet)

    # -- lowering ---------------------------------------------------------------
    @staticmethod
    def lower(dg: DGraph, hooks: Sequence, pool: Optional[int]) -> Optional['CompiledPipeline']:
        """The pipeline for the longest lowerable prefix of ``hooks`` (None: nothing to lower).  ``pool``: None = sets recycled
        only when dead (fresh-tensor semantics), R > 0 = a ring of R sets, 0 = do not lower."""
        if (pool is not None and pool <= 0) or dg.device.type != 'cuda' or not hasattr(dg, '_storage'):
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
                # (edge features by id: the post block writes the list's feature rows straight from the resident store,
                # tgmx_tgn_edge_list_by_id -- the dense [S, k, D] copies are never made)
                if i < len(hooks) and type(hooks[i]) is SampledEdgeListHook and hooks[i]._id is None and hooks[i].hop < len(nbr._num_nbrs):
                    edges = hooks[i]
                    i += 1
        return CompiledPipeline(dg, shard, neg, nbr, i, pool, dedup, edges)

    def _bind(self) -> None:
        """(Re)build the native argument block from the hooks' current state."""
        nbr, arr = self._nbr, self._arr
        nbr._ensure_state(self._dg, self._device)
        if getattr(nbr, '_bound_store', None) is not getattr(self._dg, '_storage', None):
            nbr._refresh_ts_bound(self._dg)  # a hook shared with a loader over another store: the promise below must be THIS store's
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
        p.step.guard_seed_errors = 1 if nbr._validate == 'sync' else 0
        p.step.ts_bound, p.step.sorted_ts = nbr._store_promise  # the lowered chain's batches are slices of the resident store
        self._pipe, self._step_ref = p, nbr._step
        self._scratch_ptr = nbr._step.scratch

    # -- output sets -----------------------------------------------------------------
    def _share(self, n: int) -> int:
        lo, hi = shard_bounds(n, *((self._shard.rank, self._shard.world_size) if self._shard is not None else (0, 1)))
        return hi - lo

    def _layout(self, cap: int) -> tuple:
        """Byte offsets of every sub-buffer of an output set for batches of up to ``cap`` edges (256-byte aligned, like torch's own
        allocations).  Row r of a [S, k, ...] output lives at the same offset whatever the batch size, so the delta-write state
        (tgmx_recency_step_t.out_valid: one span per row) stays valid when the batch size changes."""
        nbr = self._nbr
        D = nbr._edge_x_dim
        world = self._shard.world_size if self._shard is not None else 1
        share = -(-cap // world)
        S0 = share * len(self._roles)
        off: Dict[str, int] = {}
        pos = 0

        def take(name: str, nbytes: int) -> None:
            nonlocal pos
            off[name] = pos
            pos = (pos + nbytes + 255) & ~255

        take('neg', 4 * share)
        take('neg_time', 8 * share)
        take('seeds', 4 * S0)
        take('seed_t', 8 * S0)
        take('arange', 8 * S0)
        S = S0
        total_ids = 0
        for hop, k in enumerate(nbr._num_nbrs):
            take(f'nid{hop}', 4 * S * k)
            take(f'nts{hop}', 8 * S * k)
            take(f'nx{hop}', 0 if nbr._by_id else 4 * S * k * D)
            take(f'eid{hop}', 4 * S * k if nbr._by_id else 0)
            take(f'nv{hop}', 4 * S)  # [S] int32: the rows' spans (delta feature writes)
            take(f'nvp{hop}', 4 * S)  # the spans before the last call (byte accounting of a timed launch)
            total_ids += S * k
            S *= k
        if self._dedup is not None:
            N = int(self._dg._storage.num_nodes_global)
            extra = set(self._dedup.seed_keys or [])
            total = 2 * cap + (share if 'neg' in extra else 0) + (total_ids if 'nbr_nids' in extra else 0)
            take('uniq', 4 * max(min(total, N), 1))
            take('dev_sizes', 24)
            if self._edges is not None:
                h = self._edges.hop
                Sh = S0
                for k in nbr._num_nbrs[:h]:
                    Sh *= k
                ecap = max(Sh * nbr._num_nbrs[h], 1)
                take('ei', 16 * ecap)
                take('et', 8 * ecap)
                take('ex', 4 * ecap * D)
                take('ro', 8 * (Sh + 1))
        return off, pos, S0

    def _new_set(self, cap: int) -> _OutputSet:
        os_ = _OutputSet()
        os_.lib = self._lib
        os_.cap = cap
        os_.off, total, S0 = self._layout(cap)
        # zeros: the feature rows start as all-pad rows with span 0 (delta writes), dev_sizes as "no error"
        os_.buf = torch.zeros(max(total, 256), dtype=torch.uint8, device=self._device)
        os_.views = {}
        os_.pin, os_.event, os_.stream = None, None, None
        if self._dedup is not None:
            os_.pin = torch.zeros(3, dtype=torch.int64).pin_memory()
            ev = ctypes.c_void_p()
            _native.check(self._lib.tgmx_event_create(ctypes.byref(ev)), 'tgmx_event_create')
            os_.event = ev.value
            if self._dedup_ws is None:
                need = int(self._lib.tgmx_unique_ids_workspace_bytes(int(self._dg._storage.num_nodes_global)))
                self._dedup_ws = torch.zeros(need, dtype=torch.uint8, device=self._device)  # zeros: the bitmap cleans itself
        os_.arange_n = S0
        os_.storage = os_.buf.untyped_storage()
        self._init_set(os_)
        os_.uses = _storage_uses(os_.storage._cdata)
        return os_

    def _init_set(self, os_: _OutputSet) -> None:
        """Contents a set must hold before a batch is written into it: the seed-mask arange (never rewritten by a kernel)."""
        a = os_.off['arange']
        if os_.arange_n:
            torch.arange(os_.arange_n, out=os_.buf[a:a + 8 * os_.arange_n].view(torch.int64))
        os_.version = os_.buf._version

    def _reinit_set(self, os_: _OutputSet) -> None:
        """Somebody modified a handed-out tensor in place (the buffer's version counter moved): back to all-pad rows."""
        os_.buf.zero_()
        self._init_set(os_)

    def _views(self, os_: _OutputSet, n: int) -> _Slot:
        """The tensor set of an `n`-edge batch over output set `os_` (cached per n: views cost ~1 us each to build)."""
        sl = os_.views.get(n)
        if sl is not None:
            return sl
        if len(os_.views) >= 4:  # time-unit batching: nearly every batch has its own size -- keep the most recent few
            os_.views.pop(next(iter(os_.views)))
        nbr, buf, off = self._nbr, os_.buf, os_.off
        share = self._share(n)
        D = nbr._edge_x_dim
        S0 = share * len(self._roles)
        watch: list = []

        def view(name: str, dtype, *shape):
            numel = 1
            for d in shape:
                numel *= d
            nbytes = numel * torch.empty(0, dtype=dtype).element_size()
            t = buf[off[name]:off[name] + nbytes].view(dtype).view(*shape)
            watch.append(t)
            return t

        sl = _Slot()
        out = _native.PipelineOut()
        out.timed_hop = -1
        attrs: dict = {}
        if self._neg is not None:
            neg_t, negt_t = view('neg', torch.int32, share), view('neg_time', torch.int64, share)
            out.neg, out.neg_time = neg_t.data_ptr(), negt_t.data_ptr()
            attrs['neg'], attrs['neg_time'] = neg_t, negt_t
        seeds, seed_t = view('seeds', torch.int32, S0), view('seed_t', torch.int64, S0)
        out.seed_nid0, out.seed_ts0 = seeds.data_ptr(), seed_t.data_ptr()
        seed_n, seed_ts, nbr_n, nbr_t, nbr_x, valid = [], [], [], [], [], []
        cur_n, cur_t, S = seeds, seed_t, S0
        for hop, k in enumerate(nbr._num_nbrs):
            nid, nts = view(f'nid{hop}', torch.int32, S, k), view(f'nts{hop}', torch.int64, S, k)
            by_id = nbr._by_id and D > 0
            if by_id:  # edge ids instead of feature rows (tgmx_recency_step_t.out_eid)
                nx = view(f'eid{hop}', torch.int32, S, k)
                out.out_eid[hop] = nx.data_ptr()
            else:
                nx = view(f'nx{hop}', torch.float32, S, k, D)
            if self._delta and not by_id:
                # persistent buffers: the lookups write a feature row only from the first slot that changes (its valid slots are the
                # right-aligned tail, the rest is zero and stays zero) -- tgmx_recency_step_t.out_valid
                o, op = off[f'nv{hop}'], off[f'nvp{hop}']
                nv = (buf[o:o + 4 * S].view(torch.int32), buf[op:op + 4 * S].view(torch.int32))
                out.out_valid[hop], out.out_valid_prev[hop] = nv[0].data_ptr(), nv[1].data_ptr()
                valid.append(nv)
            out.out_nid[hop], out.out_ts[hop], out.out_x[hop] = nid.data_ptr(), nts.data_ptr(), (0 if by_id else nx.data_ptr())
            seed_n.append(cur_n)
            seed_ts.append(cur_t)
            nbr_n.append(nid)
            nbr_t.append(nts)
            nbr_x.append(nx)
            if hop + 1 < len(nbr._num_nbrs):
                cur_n, cur_t = nid.view(-1), nts.view(-1)
                watch += [cur_n, cur_t]
            S *= k
        whole = view('arange', torch.int64, S0)
        mask = {}
        for g, key in enumerate(nbr._seed_nodes_keys):
            mask[key] = whole.narrow(0, g * share, share)
            watch.append(mask[key])
        attrs.update(seed_nids=seed_n, seed_times=seed_ts, nbr_nids=nbr_n, nbr_edge_time=nbr_t, nbr_edge_x=nbr_x, seed_node_nbr_mask=mask)
        sl.out, sl.attrs, sl.nbr_nids, sl.valid = out, attrs, nbr_n, valid
        sl.post = sl.post_bufs = None
        if self._dedup is not None:
            sl.post, sl.post_bufs = self._make_post(os_, n, share, nbr_n)
        sl.watch, sl.base = watch, None  # the baseline is taken by the caller, once this frame's locals are gone
        os_.views[n] = sl
        return sl

    def _baseline(self, os_: _OutputSet, sl: _Slot) -> None:
        """Reference counts of a tensor set that nothing outside the pipeline refers to (it was just built)."""
        sl.base = _snapshot(sl.watch)
        os_.uses = _storage_uses(os_.storage._cdata)  # the new views alias the buffer

    def _make_post(self, os_: _OutputSet, n: int, share: int, nbr_n):
        """tgmx_pipeline_post_t of one tensor set: unique-id and edge-list buffers, the size mirror in pinned memory, its event."""
        nbr, buf, off = self._nbr, os_.buf, os_.off
        N = int(self._dg._storage.num_nodes_global)
        extra = set(self._dedup.seed_keys or [])
        total = 2 * n + (share if 'neg' in extra else 0) + (sum(t.numel() for t in nbr_n) if 'nbr_nids' in extra else 0)
        ucap = max(min(total, N), 1)
        uniq = buf[off['uniq']:off['uniq'] + 4 * ucap].view(torch.int32)
        dev_sizes = buf[off['dev_sizes']:off['dev_sizes'] + 24].view(torch.int64)
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
            ei = buf[off['ei']:off['ei'] + 16 * cap].view(torch.int64).view(2, cap)
            et = buf[off['et']:off['et'] + 8 * cap].view(torch.int64)
            ex = buf[off['ex']:off['ex'] + 4 * cap * D].view(torch.float32).view(cap, D)
            ro = buf[off['ro']:off['ro'] + 8 * (S + 1)].view(torch.int64)
            post.edge_hop, post.edge_cap = h, cap
            post.row_off, post.edge_index, post.edge_t, post.edge_x = ro.data_ptr(), ei.data_ptr(), et.data_ptr(), ex.data_ptr()
        post.dev_sizes, post.host_sizes, post.sizes_ready = dev_sizes.data_ptr(), os_.pin.data_ptr(), os_.event
        return post, (uniq, dev_sizes, os_.pin, os_.event, ei, et, ex, ro, N)

    def _is_free(self, os_: _OutputSet, any_stream: bool = False) -> bool:
        """Can nothing outside the pipeline reach a tensor of this set any more?"""
        if not any_stream and getattr(os_, 'stream', None) not in (None, _native.stream_ptr(self._device.index)):
            return False  # filled on another stream: "no references left" says nothing about the order of the two streams' work
        if _storage_uses(os_.storage._cdata) != os_.uses:
            return False  # a view, a detach() or any other alias of the buffer is alive
        for sl in os_.views.values():
            if sl.base is not None and _snapshot(sl.watch) != sl.base:
                return False  # a Python reference to a handed-out tensor, or autograd saved it
        return True

    def _acquire(self, n: int) -> _OutputSet:
        """The output set the next batch of `n` edges is written into."""
        if n > self._cap:
            # larger batches than any before: new sets (the old ones die with their last batch)
            self._cap = n if not self._cap else max(n, self._cap + self._cap // 2)
            self._sets = []
        sets = self._sets
        if not self._safe:
            while len(sets) < self._R:
                sets.append(self._new_set(self._cap))
            turn = self._turn
            self._turn = turn + 1
            os_ = sets[turn % self._R]
        else:
            os_ = None
            cur = _native.stream_ptr(self._device.index)
            if any(c.stream not in (None, cur) for c in sets):
                # sets filled on another stream are not recycled here (see _is_free); the dead ones leave the pool so that it can refill
                sets[:] = [c for c in sets if c.stream in (None, cur) or not self._is_free(c, any_stream=True)]
                self._last = 0
            # the set used last comes first: a consumer that drops batch i before asking for batch i + 1 stays on ONE set, whose
            # bytes are then still in the Infinity Cache
            for i in range(len(sets)):
                cand = sets[(self._last + i) % len(sets)]
                if self._is_free(cand):
                    os_, self._last = cand, (self._last + i) % len(sets)
                    break
            if os_ is None:
                os_ = self._new_set(self._cap)
                if len(sets) < _MAX_AUTO_SETS:
                    sets.append(os_)
                    self._last = len(sets) - 1
                # else: nobody keeps it -- the batch owns plain fresh tensors
        if os_.buf._version != os_.version:
            self._order_behind_worker()  # (a torch op from this thread on the loader's stream: behind the worker's pending steps)
            self._reinit_set(os_)
        return os_

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
            dedup._publish(batch, uniq.narrow(0, 0, cnt))  # (narrow: a third of the host time of a slice expression)
            if edges is not None:
                edges.add_batch_attribute(batch, 'sampled_edge_index', ei.narrow(1, 0, E))
                edges.add_batch_attribute(batch, 'sampled_edge_time', et.narrow(0, 0, E))
                edges.add_batch_attribute(batch, 'sampled_edge_x', ex.narrow(0, 0, E))

        batch._defer(finish, dedup.produces | (edges.produces if edges is not None else set()))

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
                self._order_behind_worker()
                return False
        elif n == 0:
            self._order_behind_worker()
            return False
        if nbr._validate == 'sync':
            self._validate_static(lo, n, (lo + s_lo, lo + s_hi) if shard is not None else (lo, lo + n))
        os_ = self._acquire(n)
        os_.stream = _native.stream_ptr(self._device.index)
        slot = self._views(os_, n)
        if slot.base is None:
            self._baseline(os_, slot)
        pipe = self._pipe
        ring_mode = nbr._mode == 'ring'
        if ring_mode:
            if n > nbr._scratch_edges:
                self._order_behind_worker()  # the reallocation frees scratch the worker's pending steps may still point at
                nbr._ensure_scratch(n, self._device)
            if nbr._step.scratch != self._scratch_ptr:
                self._scratch_ptr = pipe.step.scratch = nbr._step.scratch
        else:
            if nbr._csr is None or nbr._csr_store is not self._dg._storage or nbr._csr.device != nbr._device:
                self._order_behind_worker()  # an index (re)build enqueues from this thread
            nbr._ensure_csr(self._dg, lo)
            nbr._check_csr_boundary(lo)
            if nbr._epoch_lo is None:
                nbr._epoch_lo = lo
            st = pipe.step
            st.indptr, st.ring, st.ring_x, st.ev_lo = nbr._step.indptr, nbr._step.ring, nbr._step.ring_x, nbr._epoch_lo
            st.csr_cursor, st.csr_x_by_pos = nbr._step.csr_cursor, nbr._step.csr_x_by_pos
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
        asy = self._async
        if asy is not None and timer is None:
            # DGDataLoader(side_stream=True): the library's launch worker issues this step (argument blocks copied now) on the current
            # -- the loader's own -- stream, after `asy[1]` and in front of `asy[2]`; the loader collects the ticket
            tk = ctypes.c_uint64()
            rc = self._lib.tgmx_worker_pipeline_step(asy[0], pipe, lo, n, call, out, slot.post, _native.stream_ptr(self._device.index), asy[1], asy[2],
                                                     ctypes.byref(tk))
            if rc:
                _native.check(rc, 'tgmx_worker_pipeline_step')
            batch.__dict__['_ticket'] = self._last_ticket = tk.value
        else:
            stream = _native.stream_ptr(self._device.index)
            # (a timed launch under side_stream: issued from this thread -- behind every job the worker still holds, and behind the hazard event)
            self._order_behind_worker()
            rc = self._lib.tgmx_pipeline_step(pipe, lo, n, call, out, slot.post, stream)
            if rc:
                _native.check(rc, 'tgmx_pipeline_step')
        if timer is not None:
            out.timed_hop = -1
            self._log_timed(timer, slot)
        if nbr._validate == 'sync':
            self._since_check += 1
            if not self._static_ok[0] or self._since_check >= _STATUS_READ_EVERY:
                # seeds that the store does not vouch for: one device -> host read per batch.  Seeds it does vouch for: the status word
                # still carries the non-seed bits (TGMX_ST_TS_BOUND, TGMX_ST_SCRATCH) -- read it once in a while so that they surface
                self._since_check = 0
                if asy is not None and self._last_ticket:  # the status word describes THIS batch only once the worker has issued its step
                    _native.check(self._lib.tgmx_worker_wait(asy[0], self._last_ticket), 'tgmx_worker_wait')
                nbr.check()
        d = batch.__dict__
        if shard is not None:
            arr = self._arr
            d['shard_src'] = arr.src.narrow(0, lo + s_lo, s_hi - s_lo)
            d['shard_dst'] = arr.dst.narrow(0, lo + s_lo, s_hi - s_lo)
            d['shard_time'] = arr.ts.narrow(0, lo + s_lo, s_hi - s_lo)
            d['shard_lo'] = s_lo
        from .core.lazy import EdgeFeaturesById, SampledHops, SamplerCallTag

        if nbr._by_id and nbr._edge_x_dim:
            d_eids = slot.attrs['nbr_edge_x']
        # the per-hop lists carry the call's tag (tgm_amd.nn.TGAT recognises hops sampled for one another by it); the batch owns its
        # containers (a consumer may append to / reorder them), the tensors inside are the set's views
        tag = SamplerCallTag(slot.attrs['nbr_nids'], slot.attrs['nbr_edge_time'], slot.attrs.get('nbr_edge_x'))  # (by id: the edge ids)
        for key, v in slot.attrs.items():
            if type(v) is list:
                d[key] = SampledHops(v, tag) if key in _TAGGED else (list(v) if self._safe else v)
            else:
                d[key] = dict(v) if (self._safe and type(v) is dict) else v
        if nbr._by_id and nbr._edge_x_dim:
            d['nbr_edge_x'] = EdgeFeaturesById(d_eids, self._arr.edge_x, tag)  # per batch: it caches what it materializes
        if slot.post is not None:
            self._defer_post(batch, slot)
        return True

    def _order_behind_worker(self) -> None:
        """DGDataLoader(side_stream=True): work THIS thread is about to enqueue on the loader's stream (a timed step, or the hooks of a batch
        the lowered call leaves to them) goes behind every step the launch worker has been handed -- they may not have reached the stream
        yet -- and behind the consumer's reads of the output set about to be rewritten (the hazard event)."""
        asy = self._async
        if asy is None:
            return
        if self._last_ticket:
            _native.check(self._lib.tgmx_worker_wait(asy[0], self._last_ticket), 'tgmx_worker_wait')
        if asy[1]:
            _native.check(self._lib.tgmx_stream_wait_event(_native.stream_ptr(self._device.index), asy[1]), 'tgmx_stream_wait_event')

    # -- validate='sync' without a device read per batch ------------------------------------------------
    def _validate_static(self, lo: int, n: int, seed_range: tuple) -> None:
        """The reference validates every call's seeds before it touches its state (recency.py:214-229).  The lowered chain's seeds are
        rows of the RESIDENT store (+ negatives drawn in [low, high)), so their validity is a property of the store: it is established
        once, on the device, and a batch is then checked on the host against the (normally empty) sorted lists of offending edges --
        raise-per-call without a device -> host read per batch.  Seeds the store cannot vouch for (a negative range outside [0, N))
        keep the per-call read of the device status word."""
        ok = self._static_ok
        if ok is None:
            ok = self._static_ok = self._build_static()
        if not ok[0]:
            return
        import numpy as np

        _, bad_seed, bad_time, bad_edge = ok
        N = self._nbr._num_nodes
        for arr_, rng, msg in ((bad_seed, seed_range, f'Seed nodes must satisfy 0 <= x < {N}'), (bad_time, seed_range, 'Seed times must be >= 0'),
                               (bad_edge, (lo, lo + n), f'Batch edge endpoints must satisfy 0 <= x < {N}')):
            if len(arr_):
                i = int(np.searchsorted(arr_, rng[0]))
                if i < len(arr_) and arr_[i] < rng[1]:
                    raise ValueError(msg)

    def _build_static(self) -> tuple:
        nbr, arr = self._nbr, self._arr
        N = nbr._num_nodes
        if self._neg is not None and not (0 <= self._neg.low and self._neg.high <= N):
            return (False,)
        with torch.cuda.device(self._device):
            oob = lambda t: (t < 0) | (t >= N)
            bad_src, bad_dst = oob(arr.src), oob(arr.dst)
            seed = torch.zeros_like(bad_src)
            if _native.SEED_SRC in self._roles:
                seed |= bad_src
            if _native.SEED_DST in self._roles:
                seed |= bad_dst
            idx = lambda m: m.nonzero().view(-1).cpu().numpy()
            edge = (bad_src | bad_dst) if nbr._mode == 'ring' else torch.zeros_like(bad_src)
            return (True, idx(seed), idx(arr.ts < 0), idx(edge))

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

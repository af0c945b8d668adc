"""``DGDataLoader`` -- iterate a ``DGraph`` in event- or time-unit batches.

Same constructor and behaviour as tgm/data/loader.py:64-170 (batch_size,
batch_unit, on_empty, hook_manager, ``**kwargs`` forwarded to
``torch.utils.data.DataLoader``, of which it is a subclass like the reference:
``isinstance``, ``len``, ``.dataset`` = the range of slice starts, ``collate_fn``
= the loader itself).  Iteration does not go through the DataLoader machinery
(sampler, fetcher, collate indirection): a batch here is two integers plus
zero-copy views, and that machinery would be the dominant per-batch cost.
Like the reference it always runs in the caller's process -- ``num_workers > 0``
would fork the hooks' state and the device context, and is refused.

``output_pool`` (ours): the lowerable prefix of the hook chain -- [shard] ->
[negatives] -> recency sampler [-> dedup -> edge list] -- runs as ONE native
call per batch writing into persistent output sets (``tgm_amd/pipeline.py``).
``None`` (default): lowered whenever the chain allows it, with FRESH-TENSOR
semantics -- a set is reused only once nothing can reach its tensors any more.
``R > 0``: a ring of R sets recycled after R further batches (read-only
tensors).  ``0``: never lower; hooks run one by one and allocate per batch.
"""
from __future__ import annotations

import os
from typing import Any, Iterator, Literal, Optional

import torch.utils.data

from ..core import DGBatch, DGraph, TimeDeltaDG
from ..exceptions import EmptyBatchError, EventOrderedConversionError, InvalidDiscretizationError


_SIDE_STREAMS: dict = {}


def _shared_side_stream(dev):
    """ONE loader stream per device for every ``DGDataLoader(side_stream=True)`` of the process.  A stream per loader looked harmless and was
    not: the runtime maps streams onto a handful of hardware queues (4 by default), the process already owns three (the caller's, the
    library's side streams of ``tgmx_tgn_step`` and of the large ring update), and the second, third, ... loader's stream landed on a queue one
    of those uses -- its launches then serialize with them (cfg 3: 140 us per batch through the first loader of a process, 180 / 199 alternating
    through every later one, the one-stream figure; ``HISTORY.md`` 3.5).  Loaders that iterate at the same time share the stream: correct
    (stream order), merely not concurrent with each other.  TGMX_LOADER_OWN_STREAM=1: a stream per loader, as before (A/B)."""
    import torch

    if os.environ.get('TGMX_LOADER_OWN_STREAM') == '1':
        return torch.cuda.Stream(device=dev)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return s


class DGDataLoader(torch.utils.data.DataLoader):
    def __init__(
        self,
        dg: DGraph,
        batch_size: int = 1,
        batch_unit: str = 'r',
        on_empty: Literal['skip', 'raise', None] = 'skip',
        hook_manager: Optional[Any] = None,
        output_pool: Optional[int] = None,
        prefetch: int = 0,
        batch_shard: Optional[tuple] = None,
        shard_even: bool = False,
        side_stream: bool = False,
        **kwargs: Any,
    ) -> None:
        if batch_size <= 0:
            raise ValueError(f'batch_size must be > 0 but got {batch_size}')
        if on_empty not in ('skip', 'raise', None):
            raise ValueError(f"Invalid on_empty={on_empty}, expected one of: ['skip', 'raise', None]")

        unit = TimeDeltaDG(batch_unit)
        if dg.time_delta.is_event_ordered and unit.is_time_ordered:
            raise EventOrderedConversionError('Cannot iterate event-ordered dg using time-ordered batch_unit')
        if dg.time_delta.is_time_ordered and unit.is_time_ordered:
            unit = TimeDeltaDG(batch_unit, value=batch_size)
            if dg.time_delta.is_coarser_than(unit):
                raise InvalidDiscretizationError(
                    f'DGraph time delta {dg.time_delta} is strictly coarser than batch_unit={batch_unit}, '
                    f'batch_size={batch_size}; choose a larger batch or iterate event-ordered.'
                )
            batch_size = int(unit.convert(dg.time_delta))

        assert dg.start_time is not None and dg.end_time is not None
        self._dg = dg
        self._batch_size = batch_size
        self._hook_manager = hook_manager
        self._on_empty = on_empty
        self._output_pool = None if output_pool is None else int(output_pool)
        if kwargs.get('num_workers', 0):
            raise ValueError('DGDataLoader runs in the caller\'s process (hook state and device memory cannot be forked): num_workers must be 0')
        # prefetch=p (ours): iteration runs p batches ahead -- batch i is handed out after batch i + p has been ENQUEUED, so
        # hooks that must learn an output size from the device (DeduplicationHook, SampledEdgeListHook) find it waiting instead
        # of stalling the stream.  Hook state (sampler rings) does not depend on what the consumer does with a batch.
        self._prefetch = int(prefetch)
        if not 0 <= self._prefetch <= 2:
            raise ValueError(f'prefetch must be 0, 1 or 2 (hooks keep 4 size mirrors in flight), got {prefetch}')
        if self._prefetch and self._output_pool is not None and 0 < self._output_pool <= self._prefetch:
            raise ValueError(f'prefetch={prefetch} keeps {prefetch + 1} batches alive: output_pool must be 0 (fresh tensors) or > prefetch')
        # side_stream=True (ours; needs prefetch >= 1 and a recycled pool: output_pool > prefetch): the hooks' work for batch i + p is
        # enqueued on a stream of the loader's own while the consumer's work for batch i runs on the caller's stream -- the sampler /
        # dedup / edge-list chain of a TGN batch is a dozen small latency-bound launches that fit beside the model's.
        # Ordering is by events, never by the host: the consumer's stream waits for its batch's production; the production that
        # rewrites an output set waits for everything the consumer had enqueued when it was asked for (which covers its reads of
        # that set: the pool is deeper than the look-ahead).  Hook state lives on the side stream for the whole pass; the pass ends
        # by handing everything back to the caller's stream.
        self._side_stream = bool(side_stream)
        if self._side_stream and not (self._prefetch >= 1 and self._output_pool is not None and self._output_pool > self._prefetch):
            raise ValueError('side_stream=True needs prefetch >= 1 and output_pool > prefetch (the sets the consumer reads and the set being '
                             'written must be different ones)')
        self._side = None  # (torch.cuda.Stream, [events])
        self._compiled = None  # (hook list identity, CompiledPipeline or None)
        self._event_fast = False

        if unit.is_event_ordered:
            self._slice_op = dg.slice_events
            start, stop = 0, dg.num_events
            # an event-ordered view over a sub-range starts at its own first event
            lb, _ = dg._event_range
            start, stop = lb, lb + dg.num_events
            # every event is an edge and the view has no time bounds: a batch is the edge range [s, min(s + bs, stop))
            st, sl = dg._storage, dg._slice
            self._event_fast = st.num_edges == st.num_events and sl.start_time is None and sl.end_time is None
            self._stop = stop
        else:
            self._slice_op = dg.slice_time
            start, stop = dg.start_time, dg.end_time + 1
        if kwargs.get('drop_last', False):
            stop = stop - batch_size
        # batch_shard=(rank, world) (ours; SURVEY.md 8(e), tgm/data/loader.py:147-156: the loader is a range of slice starts): this
        # loader yields batches rank, rank + world, ... of the schedule -- every batch exactly the one the unsharded loader yields at that
        # position, so the ranks' outputs INTERLEAVED are the single-process batch sequence.  Only for hooks without per-batch state:
        # the recency sampler over the static index (mode='csr'), whose batches are independent units; streaming rings are refused.
        self._batch_shard = None
        self._first_start = start
        self._neg_base: dict = {}
        if batch_shard is not None:
            if not unit.is_event_ordered:
                raise ValueError('batch_shard needs event-ordered batches (batch_unit="r"): the position of a batch in the schedule is what the ranks split')
            rank, world = (int(v) for v in batch_shard)
            if not 0 <= rank < world:
                raise ValueError(f'batch_shard: rank {rank} outside [0, {world})')
            self._batch_shard = (rank, world)
            # The schedule's batch count is rarely a multiple of ``world``: ranks then yield ceil or floor(total / world) batches and
            # ``len(loader)`` differs per rank.  A loop with a per-step collective (DDP's gradient all-reduce, TGNMemory(shard_commits=True)'s
            # all-gather) would hang at the end of the epoch: ``shard_even=True`` stops EVERY rank after floor(total / world) batches (the
            # schedule's last total % world batches are dropped, like ``drop_last``); without it wrap the loop in
            # ``torch.distributed.algorithms.Join`` or keep it collective-free.
            total = len(range(start, stop, batch_size))
            hi = start + (total // world) * world * batch_size if shard_even else stop
            self._starts = range(start + rank * batch_size, min(stop, hi), batch_size * world)
        else:
            self._starts = range(start, stop, batch_size)
        # the reference's base-class call (loader.py:147-149): torch validates the keyword arguments; unknown ones raise TypeError
        super().__init__(self._starts, 1, shuffle=False, collate_fn=self, **kwargs)

    def __del__(self) -> None:
        side = getattr(self, '_side', None)
        if side is not None and len(side) > 2 and side[2]:
            try:
                from .. import _native

                lib = _native.load()
                lib.tgmx_worker_destroy(side[2])  # finishes pending jobs, joins the library's launch thread
                for e in side[1]:
                    lib.tgmx_event_destroy(e)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass
            self._side = None

    @property
    def dgraph(self) -> DGraph:
        return self._dg

    def __len__(self) -> int:
        return len(self._starts)

    def _check_shardable(self, hooks) -> None:
        for h in hooks:
            if getattr(h, 'has_state', False) and not getattr(h, 'batches_are_independent', False):
                raise ValueError(
                    f'batch_shard: {type(h).__name__} carries state from batch to batch, so a rank cannot skip the batches of the other '
                    "ranks.  The recency sampler is stateless over the static index: RecencyNeighborHook(mode='csr', batch_size=...)."
                )

    def __call__(self, slice_start, _deferred: bool = False) -> DGBatch:
        """Materialize the batch beginning at ``slice_start`` and run the active hooks."""
        s = slice_start[0] if isinstance(slice_start, (list, tuple)) else slice_start
        if self._batch_shard is not None and self._hook_manager is not None and hasattr(self._hook_manager, 'active_hooks'):
            hooks = self._hook_manager.active_hooks()
            if getattr(self, '_shard_checked', None) is not hooks:
                self._check_shardable(hooks)
                self._shard_checked = hooks
            self._anchor_epoch(hooks)
            # generated negatives are a function of (seed, call number, position): the call number of schedule position j is the one
            # the unsharded loader would be at, so every rank draws exactly the negatives of the batches it takes
            j = (s - self._first_start) // self._batch_size
            for h in hooks:
                if hasattr(h, '_rng_seed') and hasattr(h, '_calls'):
                    base = self._neg_base.get(id(h))
                    if base is None:
                        base = self._neg_base[id(h)] = h._calls
                    h._calls = base + j
        if self._output_pool != 0 and hasattr(self._hook_manager, 'active_hooks'):
            batch = self._call_compiled(s, _deferred)
            if batch is not None:
                return batch if _deferred else batch._finalize()
        view = self._slice_op(s, s + self._batch_size)
        batch = view.materialize()
        hm = self._hook_manager
        if hm is not None:
            batch.__dict__['_deferred'] = _deferred
            if hasattr(hm, 'active_hooks'):
                for h in hm.active_hooks():
                    if '_pending' in batch.__dict__:
                        self._settle_for(batch, h)
                    batch = h(view, batch)
            else:  # a foreign manager (the reference's): its own entry point
                batch = hm.execute_active_hooks(view, batch)
        return batch if _deferred else batch._finalize()

    def _call_compiled(self, s: int, deferred: bool = False) -> Optional[DGBatch]:
        """The batch through the lowered hook chain (None: nothing lowerable / this batch is left to the hooks)."""
        hm = self._hook_manager
        hooks = hm.active_hooks()
        cached = self._compiled
        if cached is None or cached[0] is not hooks:
            from ..pipeline import CompiledPipeline

            cached = self._compiled = (hooks, CompiledPipeline.lower(self._dg, hooks, self._output_pool))
        pipe = cached[1]
        if pipe is None:
            return None
        dg = self._dg
        view = None
        if self._event_fast:
            lo, hi = s, min(s + self._batch_size, self._stop)
            lb = lo
        else:
            view = self._slice_op(s, s + self._batch_size)
            lo, hi = view._edge_range
            lb = view._event_range[0]
        n = hi - lo
        arr = dg._storage.on(dg._device)
        if view is None or (arr.node_x is None and arr.node_y is None):
            batch = DGBatch(arr.src.narrow(0, lo, n), arr.dst.narrow(0, lo, n), arr.ts.narrow(0, lo, n))
            batch._edge_lo, batch._event_lo = lo, lb
            if n > 0:
                if arr.edge_x is not None:
                    batch.edge_x = arr.edge_x.narrow(0, lo, n)
                if arr.edge_type is not None:
                    batch.edge_type = arr.edge_type.narrow(0, lo, n)
        else:
            batch = view.materialize()
        batch.__dict__['_deferred'] = deferred
        if not pipe.step(lo, n, batch):
            batch.__dict__['_deferred'] = False
            return None
        rest = hooks[pipe.n_lowered :] if pipe.n_lowered < len(hooks) else None
        if rest:
            if view is None:
                view = self._slice_op(s, s + self._batch_size)
            for h in rest:
                if '_pending' in batch.__dict__:
                    self._settle_for(batch, h)
                batch = h(view, batch)
        return batch

    def _anchor_epoch(self, hooks) -> None:
        """A static-index sampler anchors its epoch (the first edge whose history is visible) at the first batch it sees after
        ``reset_state``; rank r of a batch-sharded loader first sees batch r, so the anchor is handed over: the schedule's first edge."""
        if self._event_fast:
            first = self._first_start
        else:
            first = self._slice_op(self._first_start, self._first_start + self._batch_size)._edge_range[0]
        for h in hooks:
            anchor = getattr(h, '_anchor_epoch', None)
            if anchor is not None:
                anchor(first)

    @staticmethod
    def _settle_for(batch: DGBatch, h: Any) -> None:
        """Batch attributes whose SIZE is learnt from the device are published by finalizers that a prefetching loader runs one
        batch later; a hook that requires such an attribute gets them run before it is called."""
        req = getattr(h, 'requires', None)
        if req is not None:
            req = set(req) - set(getattr(h, '_device_side_requires', ()))
        batch._settle(req)

    @staticmethod
    def _is_batch_empty(batch: DGBatch) -> bool:
        n = batch.edge_src.numel()
        n += batch.node_x_nids.numel() if batch.node_x_nids is not None else 0
        n += batch.node_y_nids.numel() if batch.node_y_nids is not None else 0
        return n == 0

    def __iter__(self) -> Iterator[DGBatch]:
        if self._batch_shard is not None:
            yield from self._iter_sharded()
            return
        if self._prefetch > 0:
            yield from self._iter_prefetch()
            return
        for s in self._starts:
            batch = self(s)
            if self._on_empty is not None and self._is_batch_empty(batch):
                if self._on_empty == 'raise':
                    raise EmptyBatchError('Empty batch encountered')
                continue
            yield batch

    def _iter_sharded(self) -> Iterator[DGBatch]:
        """One pass over this rank's batches; afterwards the negative samplers' call counters stand where the unsharded pass leaves them
        (the next pass -- another epoch, the validation loader -- then draws what a single process would)."""
        self._neg_base = {}
        total = len(range(self._first_start, self._starts.stop, self._batch_size))
        src = self._iter_prefetch() if self._prefetch > 0 else None
        try:
            if src is not None:
                yield from src
            else:
                for s in self._starts:
                    batch = self(s)
                    if self._on_empty is not None and self._is_batch_empty(batch):
                        if self._on_empty == 'raise':
                            raise EmptyBatchError('Empty batch encountered')
                        continue
                    yield batch
        finally:
            hm = self._hook_manager
            if hm is not None and hasattr(hm, 'active_hooks'):
                for h in hm.active_hooks():
                    base = self._neg_base.get(id(h))
                    if base is not None:
                        h._calls = base + total
            self._neg_base = {}

    def _iter_prefetch(self) -> Iterator[DGBatch]:
        from collections import deque

        if self._side_stream:
            yield from self._iter_side_stream()
            return
        ahead: deque = deque()
        for s in self._starts:
            batch = self(s, _deferred=True)
            if self._on_empty is not None and self._is_batch_empty(batch):
                if self._on_empty == 'raise':
                    raise EmptyBatchError('Empty batch encountered')
                batch._finalize()
                continue
            ahead.append(batch)
            if len(ahead) > self._prefetch:
                yield ahead.popleft()._finalize()
        while ahead:
            yield ahead.popleft()._finalize()

    def _iter_side_stream(self) -> Iterator[DGBatch]:
        """``_iter_prefetch`` with the production on the loader's own stream (see ``side_stream`` in ``__init__``).  When the whole hook
        chain is lowered (one native call per batch) that call is issued by the library's launch worker (``tgmx_worker_pipeline_step``:
        a host thread of the library, no GIL), so this thread pays for the hooks' Python only, not for their launches."""
        import ctypes
        from collections import deque

        import torch

        from .. import _native

        dev = self._dg._device
        if dev.type != 'cuda':
            raise ValueError('side_stream=True needs a device-resident graph (DGraph(..., device="cuda"))')
        lib = _native.load()
        if self._side is not None and self._side[0].device != dev:
            # the graph moved to another device: the old worker (joined after its pending jobs) and events go before the new ones come
            _native.check(lib.tgmx_worker_destroy(self._side[2]), 'tgmx_worker_destroy')
            for e in self._side[1]:
                lib.tgmx_event_destroy(e)
            self._side = None
        if self._side is None:
            n_ev = 2 * (self._prefetch + 2)
            evs = []
            for _ in range(n_ev):
                e = ctypes.c_void_p()
                _native.check(lib.tgmx_event_create_sync(ctypes.byref(e)), 'tgmx_event_create_sync')
                evs.append(e)
            with torch.cuda.device(dev):
                w = ctypes.c_void_p()
                _native.check(lib.tgmx_worker_create(ctypes.byref(w)), 'tgmx_worker_create')
            self._side = (_shared_side_stream(dev), evs, w)
        side, evs, worker = self._side
        main = torch.cuda.current_stream(dev)
        side_p, main_p = side.cuda_stream, main.cuda_stream
        record, wait, handoff, wwait = lib.tgmx_event_record, lib.tgmx_stream_wait_event, lib.tgmx_stream_handoff, lib.tgmx_worker_wait
        set_stream = torch.cuda.set_stream
        n_prod = len(evs) // 2
        use_worker = os.environ.get('TGMX_LOADER_WORKER', '1') != '0'  # A/B knob: 0 = this thread issues the loader's launches itself
        finalize_first = os.environ.get('TGMX_LOADER_FINALIZE_FIRST', '1') != '0'  # A/B knob: 0 = read the sizes back behind the submission
        ahead: deque = deque()
        j = 0
        last_ticket = 0
        # the hooks' state was last touched on the caller's stream (reset_state, an earlier pass): the side stream starts behind it
        _native.check(handoff(main_p, side_p, evs[n_prod]), 'tgmx_stream_handoff')
        try:
            for s in self._starts:
                if finalize_first and len(ahead) == self._prefetch and ahead:
                    # the batch this iteration will hand out: its three sizes are read back NOW, while the launch worker is idle -- behind the
                    # submission below the read (hipEventSynchronize) queued behind the worker's launches on the runtime's locks
                    b0, _, tk0 = ahead[0]
                    if tk0 is not None:
                        _native.check(wwait(worker, tk0), 'tgmx_worker_wait')
                    b0._finalize()
                hz = None
                if j:
                    # this production may rewrite the set of a batch the consumer has finished ENQUEUING work for: after that work
                    hz = evs[n_prod + j % n_prod]
                    _native.check(record(hz, main_p), 'tgmx_event_record')
                ev = evs[j % n_prod]
                pipe = self._compiled[1] if self._compiled is not None else None
                hm = self._hook_manager
                lowered = (use_worker and pipe is not None and hm is not None and pipe.n_lowered == len(hm.active_hooks()) and pipe._shard is None
                           and self._compiled[0] is hm.active_hooks())
                if lowered:
                    pipe._async = (worker, hz, ev)  # the worker waits for hz on the side stream, issues the step, records ev
                else:
                    if last_ticket:  # earlier jobs are issued before this thread enqueues on the same stream
                        _native.check(wwait(worker, last_ticket), 'tgmx_worker_wait')
                    if hz is not None:
                        _native.check(wait(side_p, hz), 'tgmx_stream_wait_event')
                set_stream(side)
                try:
                    batch = self(s, _deferred=True)
                finally:
                    set_stream(main)
                    if lowered:
                        pipe._async = None
                ticket = batch.__dict__.pop('_ticket', None)
                if ticket is not None:
                    last_ticket = ticket
                else:
                    if lowered:
                        # the lowered call left this batch to the hooks (it submitted no job): what they enqueued from this thread is
                        # ordered by hand -- behind the worker's earlier jobs would have been too late for the hazard, so drain and accept
                        # that such batches (empty shares) are rare
                        if last_ticket:
                            _native.check(wwait(worker, last_ticket), 'tgmx_worker_wait')
                    _native.check(record(ev, side_p), 'tgmx_event_record')
                if self._on_empty is not None and self._is_batch_empty(batch):
                    if self._on_empty == 'raise':
                        raise EmptyBatchError('Empty batch encountered')
                    if ticket is not None:
                        _native.check(wwait(worker, ticket), 'tgmx_worker_wait')
                    batch._finalize()
                    continue
                j += 1
                ahead.append((batch, ev, ticket))
                if len(ahead) > self._prefetch:
                    b, e, tk = ahead.popleft()
                    if tk is not None:  # (normally issued long ago) its events must have been recorded before anybody waits on them
                        _native.check(wwait(worker, tk), 'tgmx_worker_wait')
                    _native.check(wait(main_p, e), 'tgmx_stream_wait_event')
                    yield b._finalize()
            while ahead:
                b, e, tk = ahead.popleft()
                if tk is not None:
                    _native.check(wwait(worker, tk), 'tgmx_worker_wait')
                _native.check(wait(main_p, e), 'tgmx_stream_wait_event')
                yield b._finalize()
        finally:
            # whatever comes next on the caller's stream (another pass, reset_state, hook.check()) sees the side stream's work
            if last_ticket:
                wwait(worker, last_ticket)
            handoff(side_p, main_p, evs[n_prod])

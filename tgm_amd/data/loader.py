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

from typing import Any, Iterator, Literal, Optional

import torch.utils.data

from ..core import DGBatch, DGraph, TimeDeltaDG
from ..exceptions import EmptyBatchError, EventOrderedConversionError, InvalidDiscretizationError


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

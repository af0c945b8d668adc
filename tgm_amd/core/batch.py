"""``DGBatch`` -- the mutable record hooks read from and write onto.

Mirrors tgm/core/batch.py:11-45 (same field names, order and defaults) so that
hooks written against the reference keep working, and ``dataclasses.asdict`` /
``fields`` / ``==`` see exactly the reference's record.  Two private instance
attributes (NOT dataclass fields) are set by :meth:`DGraph.materialize`:
``_edge_lo`` (index of the batch's first edge in the device-resident edge
store; the batch's edges are the contiguous range ``[_edge_lo, _edge_lo +
len(edge_src))``) and ``_event_lo`` (the slice's first global event index).
They let device hooks address the resident store by edge id instead of copying
feature rows.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, ClassVar, Optional

from torch import Tensor


@dataclass
class DGBatch:
    edge_src: Tensor
    edge_dst: Tensor
    edge_time: Tensor
    edge_x: Optional[Tensor] = None
    edge_type: Optional[Tensor] = None

    node_x_time: Optional[Tensor] = None
    node_x_nids: Optional[Tensor] = None
    node_x: Optional[Tensor] = None

    node_y_time: Optional[Tensor] = None
    node_y_nids: Optional[Tensor] = None
    node_y: Optional[Tensor] = None

    # class-level defaults of the two private instance attributes (ClassVar: not fields, so asdict(batch) is the reference's)
    _edge_lo: ClassVar[Optional[int]] = None
    _event_lo: ClassVar[Optional[int]] = None

    # Hooks whose output SIZE is only known on the device (unique ids, compacted edge lists) enqueue their kernels, start an
    # asynchronous copy of the size and register a finalizer here instead of waiting for it; the loader runs the
    # finalizers right away (default) or one batch later (DGDataLoader(prefetch=1)), when the size arrived long ago.
    def _defer(self, fn, produces=frozenset()) -> None:
        """``produces``: the batch attributes ``fn`` publishes -- a hook further down the chain that requires one of them makes the
        loader run the finalizers first (:meth:`_settle`)."""
        if self.__dict__.get('_deferred', False):
            self.__dict__.setdefault('_pending', []).append((fn, frozenset(produces)))
        else:
            fn()

    def _settle(self, requires) -> None:
        """Run the pending finalizers now if any of them publishes an attribute in ``requires`` (None: whatever they publish)."""
        pending = self.__dict__.get('_pending')
        if pending and (requires is None or any(not p or (p & requires) for _, p in pending)):
            del self.__dict__['_pending']
            for fn, _ in pending:
                fn()

    def _finalize(self) -> 'DGBatch':
        pending = self.__dict__.pop('_pending', None)
        self.__dict__['_deferred'] = False
        if pending:
            for fn, _ in pending:
                fn()
        return self

    def __str__(self) -> str:
        def describe(v: Any) -> str:
            if isinstance(v, Tensor):
                return str(list(v.shape))
            if isinstance(v, str):
                return v
            if isinstance(v, dict):
                return 'dict(' + '|'.join(sorted({describe(x) for x in v})) + f' x{len(v)})'
            if isinstance(v, (list, tuple)):
                kinds = '|'.join(sorted({describe(x) for x in v}))
                return f'{type(v).__name__}({kinds} x{len(v)})'
            return type(v).__name__

        parts = [f'{k} = {describe(v)}' for k, v in vars(self).items() if not k.startswith('_')]
        return 'DGBatch(' + ', '.join(parts) + ')'

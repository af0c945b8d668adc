"""Hook protocol and base classes (contract of tgm/hooks/base.py:10-103).

A hook is anything with ``has_state``, ``requires``, ``produces``,
``__call__(dg, batch) -> batch`` and ``reset_state()``.  ``produces`` names are
suffixed with ``_{id}`` when the hook was given an ``id``; ``requires`` never
is.  Seedable hooks add their ``seed_keys`` to ``requires``.
"""
from __future__ import annotations

from typing import Any, List, Optional, Protocol, Set, runtime_checkable

from ..core import DGBatch, DGraph


@runtime_checkable
class DGHook(Protocol):
    has_state: bool

    @property
    def requires(self) -> Set[str]: ...

    @property
    def produces(self) -> Set[str]: ...

    def __call__(self, dg: DGraph, batch: DGBatch) -> DGBatch: ...

    def reset_state(self) -> None: ...


class BaseDGHook:
    """Bookkeeping shared by every hook: requires / produces / id suffixing."""

    _cls_requires: Set[str] = set()
    _cls_produces: Set[str] = set()
    has_state: bool = False

    def __init__(self) -> None:
        self._requires: Set[str] = set()
        self._produces: Set[str] = set()
        self._id: Optional[str] = None
        self.seed_keys: Optional[List[str]] = None

    def __post_init__(self) -> None:
        """Call at the end of a subclass ``__init__`` (after ``_id`` / ``seed_keys`` are set)."""
        self._requires |= set(type(self).__dict__.get('_cls_requires', set()))
        self._produces |= set(type(self).__dict__.get('_cls_produces', set()))
        if getattr(self, 'seed_keys', None):
            self._requires |= set(self.seed_keys)  # type: ignore[arg-type]

    @property
    def requires(self) -> Set[str]:
        return self._requires

    @property
    def produces(self) -> Set[str]:
        if self._id is None:
            return self._produces
        return {f'{name}_{self._id}' for name in self._produces}

    def __repr__(self) -> str:
        name = type(self).__name__
        return f'{name}_{self._id}' if self._id else name

    def __call__(self, dg: DGraph, batch: DGBatch) -> DGBatch:  # pragma: no cover - abstract
        raise NotImplementedError

    def reset_state(self) -> None:
        pass

    def add_batch_attribute(self, batch: DGBatch, name: str, value: Any) -> None:
        setattr(batch, f'{name}_{self._id}' if self._id else name, value)


class StatelessHook(BaseDGHook):
    has_state = False


class StatefulHook(BaseDGHook):
    has_state = True


class SeedableHook(BaseDGHook):
    """Marker: the hook reads extra ``seed_keys`` attributes from the batch."""

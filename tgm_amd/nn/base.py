"""What ``HookManager.validate_requirement`` expects of a model (tgm/nn/base.py:7): a callable that names, in
``requires``, the batch attributes its forward reads.  ``missing_attributes`` is the check itself, shared with the
hook manager's error message."""
from __future__ import annotations

from typing import Any, Iterable, List, Protocol, Set, runtime_checkable


@runtime_checkable
class EncoderModule(Protocol):
    requires: Set[str]

    def __call__(self, batch: Any, *args: Any, **kwargs: Any) -> Any: ...


def missing_attributes(module: 'EncoderModule', produced: Iterable[str]) -> List[str]:
    """Attributes ``module`` needs that none of the active hooks (nor the bare batch) provides, sorted."""
    have = set(produced)
    return sorted(a for a in module.requires if a not in have)

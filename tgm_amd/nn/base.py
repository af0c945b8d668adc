"""Protocol HookManager.validate_requirement checks modules against (tgm/nn/base.py:7)."""
from __future__ import annotations

from typing import Any, Protocol, Set, runtime_checkable


@runtime_checkable
class EncoderModule(Protocol):
    requires: Set[str]

    def __call__(self, batch: Any, *args: Any, **kwargs: Any) -> Any: ...

"""Class registry used for HookManager's "did you mean ..." suggestions
(role of tgm/hooks/registry.py:8-22)."""
from __future__ import annotations

from typing import Dict, List, Type

_REGISTRY: Dict[str, Type] = {}


def hook(cls: Type) -> Type:
    """Class decorator: make ``cls`` discoverable by name."""
    _REGISTRY[cls.__name__] = cls
    return cls


def list_hooks() -> List[Type]:
    return list(_REGISTRY.values())

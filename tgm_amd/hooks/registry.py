"""Class registry used for HookManager's "did you mean ..." suggestions
(tgm/hooks/registry.py:8-22: a plain list in registration order -- registering
a class twice lists it twice, and ``list_hooks`` hands out the list itself)."""
from __future__ import annotations

from typing import List, Type

_HOOK_REGISTRY: List[Type] = []


def hook(cls: Type) -> Type:
    """Class decorator: make ``cls`` discoverable by name."""
    _HOOK_REGISTRY.append(cls)
    return cls


def list_hooks() -> List[Type]:
    return _HOOK_REGISTRY

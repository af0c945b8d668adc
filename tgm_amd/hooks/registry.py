"""Class registry behind HookManager's "did you mean ..." suggestions (role of tgm/hooks/registry.py:8-22).

Semantics the reference's tests pin (test/unit/test_hooks/test_registry.py): registration order is kept, a class registered twice is
listed twice, ``list_hooks()`` hands out the live list, ``hook`` returns the class untouched."""
from __future__ import annotations

from typing import List, Type

_registered: List[Type] = []


def hook(cls: Type) -> Type:
    """Class decorator: make ``cls`` discoverable by the attributes it produces."""
    _registered.append(cls)
    return cls


def list_hooks() -> List[Type]:
    return _registered


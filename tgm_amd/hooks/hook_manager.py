"""``HookManager`` -- keyed + shared hook lists with dependency ordering.

Behavioural contract of tgm/hooks/hook_manager.py:38-462: hooks are ordered by a
Kahn topological sort over ``produces & requires`` (ties keep registration
order), with one extra implicit edge "whatever produces ``neg`` runs before
whatever produces ``nbr_nids``" (hook_manager.py:427-430) so neighbor samplers
see the negatives as seeds.  The order is resolved lazily per key and cached
until another hook is registered.
"""
from __future__ import annotations

import difflib
from collections import deque
from contextlib import contextmanager
from typing import Any, Dict, Iterator, List, Optional, Set

from ..core import DGBatch, DGraph
from ..exceptions import BadEncoderProtocolError, BadHookProtocolError, UnresolvableHookDependenciesError
from .base import DGHook
from .registry import list_hooks

# attributes every materialized batch carries without any hook producing them
CORE_ATTRIBUTE: Set[str] = {
    'edge_src',
    'edge_dst',
    'edge_time',
    'edge_type',
    'node_x_time',
    'node_x_nids',
    'node_y_time',
    'node_y_nids',
    'node_type',
}


class HookManager:
    def __init__(self, keys: List[str]) -> None:
        if not len(keys):
            raise ValueError('HookManager keys list must be non-empty')
        self._registered_key = keys
        self._key_to_hooks: Dict[str, List[DGHook]] = {k: [] for k in keys}
        self._dirty: Dict[str, bool] = {k: False for k in keys}
        self._shared_hooks: List[DGHook] = []
        self._active_key: Optional[str] = None

    # -- registration -------------------------------------------------------
    @property
    def keys(self) -> List[str]:
        return self._registered_key

    def register_shared(self, hook: DGHook) -> None:
        self._check_hook(hook)
        self._check_idle()
        self._shared_hooks.append(hook)
        for k in self._dirty:
            self._dirty[k] = True

    def register(self, key: str, hook: DGHook) -> None:
        self._check_key(key)
        self._check_hook(hook)
        self._check_idle()
        self._key_to_hooks[key].append(hook)
        self._dirty[key] = True

    # -- activation ---------------------------------------------------------
    def set_active_hooks(self, key: str) -> None:
        self._check_key(key)
        self._active_key = key

    @contextmanager
    def activate(self, key: str) -> Iterator[None]:
        previous = self._active_key
        self.set_active_hooks(key)
        try:
            yield
        finally:
            self._active_key = previous

    # -- execution ----------------------------------------------------------
    def execute_active_hooks(self, dg: DGraph, batch: DGBatch) -> DGBatch:
        key = self._active_key
        if key is None:
            raise RuntimeError('No active key set. Use activate() context manager.')
        if self._dirty[key]:
            self.resolve_hooks(key)
        for h in self._key_to_hooks[key]:
            batch = h(dg, batch)
        return batch._finalize() if hasattr(batch, '_finalize') else batch

    def active_hooks(self) -> List[DGHook]:
        """The resolved (dependency-ordered) hook list of the active key.  The list object is replaced, not edited, when
        hooks are registered, so callers may cache on its identity."""
        key = self._active_key
        if key is None:
            raise RuntimeError('No active key set. Use activate() context manager.')
        if self._dirty[key]:
            self.resolve_hooks(key)
        return self._key_to_hooks[key]

    def reset_state(self, key: Optional[str] = None) -> None:
        if key is not None:
            self._check_key(key)
        for h in self._shared_hooks:
            h.reset_state()
        for k in [key] if key is not None else list(self._key_to_hooks):
            for h in self._key_to_hooks[k]:
                h.reset_state()

    def resolve_hooks(self, key: Optional[str] = None) -> None:
        if key is not None:
            self._check_key(key)
        for k in [key] if key else list(self._key_to_hooks):
            own = [h for h in self._key_to_hooks[k] if h not in self._shared_hooks]
            self._key_to_hooks[k] = self._topological_sort_hooks(self._shared_hooks + own)
            self._dirty[k] = False

    # -- module requirement validation ----------------------------------------
    def validate_requirement(self, module: Any, key: Optional[str] = None) -> None:
        """Check that ``module.requires`` is covered by the hooks under ``key`` (or every key)."""
        from ..nn.base import EncoderModule, missing_attributes

        if not isinstance(module, EncoderModule):
            raise BadEncoderProtocolError(
                f'Cannot validate {type(module).__name__}: must be callable and expose a `requires` attribute'
            )
        if key is not None:
            self._check_key(key)
        for k in [key] if key is not None else list(self._key_to_hooks):
            hooks = self._key_to_hooks[k] + self._shared_hooks
            if missing_attributes(module, CORE_ATTRIBUTE.union(*(h.produces for h in hooks))):
                self._explain_missing(set(module.requires), hooks, k)

    def _explain_missing(self, needed: Set[str], hooks: List[DGHook], key: str) -> None:
        available = CORE_ATTRIBUTE.union(*(h.produces for h in hooks))
        missing = needed - available
        if not missing:
            return
        msg = f"Cannot resolve the following requirements {missing} from any hook registered under key '{key}'.\nSuggestions:"
        for attr in missing:
            hinted = False
            for cls in list_hooks():
                made: Set[str] = getattr(cls, '_cls_produces', set())
                close = difflib.get_close_matches(attr, made, n=2, cutoff=0.6)
                if attr in made:
                    msg += f"\n\t- '{attr}': Found hook that produces '{attr}'. To resolve this, please register '{cls.__name__}' with key '{key}'"
                    hinted = True
                elif close:
                    names = ' or '.join(f"'{c}'" for c in close)
                    msg += (
                        f"\n\t- '{attr}': Do you mean {names}?. If so, please update the module requirement with the "
                        f"correct name and register '{cls.__name__}' with key '{key}' to resolve this."
                    )
                    hinted = True
                elif attr.lower() in (cls.__doc__ or '').lower():
                    msg += (
                        f"\n\t- '{attr}': Found keyword '{attr}' in '{cls.__name__}' documentation. If this hook produces what you "
                        f"are looking for, update the module requirement with the correct name and register '{cls.__name__}' with key '{key}'."
                    )
                    hinted = True
            if not hinted:
                msg += f"\n\t- '{attr}': Can not find any existing hooks that satisfy this requirement."
        raise UnresolvableHookDependenciesError(msg)

    # -- checks ---------------------------------------------------------------
    def _check_hook(self, hook: Any) -> None:
        if not isinstance(hook, DGHook):
            raise BadHookProtocolError(
                f'Cannot register hook {type(hook).__name__}: must implement __call__(dg, batch) -> DGBatch, '
                'reset_state(), requires and produces properties.'
            )

    def _check_idle(self) -> None:
        if self._active_key is not None:
            raise RuntimeError('Cannot register hooks while a key is active. Register hooks before using `activate`.')

    def _check_key(self, key: str) -> None:
        if key not in self._key_to_hooks:
            raise KeyError(f'{key} was not a declared key in the hook manager')

    # -- ordering ---------------------------------------------------------------
    @staticmethod
    def _topological_sort_hooks(hooks: List[DGHook]) -> List[DGHook]:
        provided = CORE_ATTRIBUTE.union(*(h.produces for h in hooks))
        unmet: Set[str] = set()
        for h in hooks:
            unmet |= h.requires - provided
        if unmet:
            raise UnresolvableHookDependenciesError(
                f'Cannot resolve hook dependencies: required attributes not produced by any hook: {unmet}'
            )

        n = len(hooks)
        succ: List[List[int]] = [[] for _ in range(n)]
        indeg = [0] * n
        for i, a in enumerate(hooks):
            for j, b in enumerate(hooks):
                if i == j:
                    continue
                if a.produces & b.requires:
                    succ[i].append(j)
                    indeg[j] += 1
                # negatives must exist before a neighbor sampler seeds from them
                if 'neg' in a.produces and 'nbr_nids' in b.produces:
                    succ[i].append(j)
                    indeg[j] += 1

        ready = deque(i for i in range(n) if indeg[i] == 0)
        order: List[int] = []
        while ready:
            i = ready.popleft()
            order.append(i)
            for j in succ[i]:
                indeg[j] -= 1
                if indeg[j] == 0:
                    ready.append(j)

        if len(order) != n:
            done = set(order)
            have = set().union(*[hooks[i].produces for i in order]) if order else set()
            msg = 'Cannot resolve hook dependencies:\n'
            for i in range(n):
                if i not in done:
                    msg += f'\n - {hooks[i]!r} requires {hooks[i].requires - have} but not produced (or stuck in cycle)'
            raise UnresolvableHookDependenciesError(msg)
        return [hooks[i] for i in order]

    def __str__(self) -> str:
        line = lambda h: f'    - {h!r} (requires={h.requires}, produces={h.produces})'
        out = ['HookManager:', '  Shared hooks:'] + [line(h) for h in self._shared_hooks]
        out += [f'  Active key: {self._active_key}', '  Keyed hooks:']
        for k, hs in self._key_to_hooks.items():
            out.append(f'    {k}:')
            out += [line(h) for h in hs]
        return '\n'.join(out)

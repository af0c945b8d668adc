"""When did a module's parameters last change?  Derived copies of the weights (padded / transposed layouts for the kernels, the
folded query side of the TGAT inference path, stacked projections) are cached against this key.

``Tensor._version`` alone is NOT enough: the fused optimizers (``torch.optim.Adam(..., fused=True)``, ``AdamW``, ``SGD``) update
the parameters with one multi-tensor kernel that does not bump it (checked on torch 2.10: ``p._version`` is unchanged after
``opt.step()``), so a cache keyed on (data_ptr, _version) would keep serving the weights of step 0 -- training that silently
never sees its own updates.  The key therefore also carries a process-wide count of optimizer steps (a global post-step hook:
any optimizer, any parameter group).  What is still invisible: writes through ``p.data`` (its own version counter) -- code that
does that calls :func:`invalidate_parameter_caches`.
"""
from typing import Iterable, Tuple, Union

import torch
from torch.nn.modules.module import register_module_module_registration_hook, register_module_parameter_registration_hook
from torch.optim.optimizer import register_optimizer_step_post_hook

_epoch = 0
_registrations = 0  # bumped whenever ANY module registers a parameter (construction, setattr, load_state_dict(assign=True))


def _after_step(optimizer, args, kwargs) -> None:
    global _epoch
    _epoch += 1


register_optimizer_step_post_hook(_after_step)


def _on_registration(module, name, param) -> None:
    global _registrations
    _registrations += 1


register_module_parameter_registration_hook(_on_registration)
# replacing a submodule (``model.attn[0] = other``) changes the parameter set without registering a parameter
register_module_module_registration_hook(_on_registration)


def param_list(module: torch.nn.Module) -> Tuple[torch.Tensor, ...]:
    """``tuple(module.parameters())``, cached on the module: walking the module tree costs ~20 us, several times per batch on the TGN
    path.  The cache is dropped when any module registers a parameter (``.to()`` / ``load_state_dict`` keep the Parameter objects)."""
    cached = module.__dict__.get('_tgmx_plist')
    if cached is None or cached[0] != _registrations:
        cached = module.__dict__['_tgmx_plist'] = (_registrations, tuple(module.parameters()))
    return cached[1]


def refresh_param_list(module: torch.nn.Module) -> Tuple[torch.Tensor, ...]:
    """Re-walk the module tree now.  For the paths that assign ``module._parameters[key]`` directly and therefore run no registration
    hook (``Module._apply`` under ``torch.__future__.set_overwrite_module_params_on_conversion(True)`` / ``set_swap_module_params_on_conversion``):
    ``TransientCaches._apply`` calls this after every ``.to()`` / ``.float()`` / ``.cuda()``."""
    module.__dict__.pop('_tgmx_plist', None)
    return param_list(module)


def invalidate_parameter_caches() -> None:
    """Drop every cached derived copy of any module's weights (after in-place writes autograd cannot see, e.g. ``p.data.mul_()``)."""
    global _epoch
    _epoch += 1


def param_key(params: Union[torch.nn.Module, Iterable[torch.Tensor]]) -> Tuple:
    if isinstance(params, torch.nn.Module):
        params = param_list(params)
    return (_epoch,) + tuple((p.data_ptr(), p._version) for p in params)


class TransientCaches:
    """Mixin (in front of ``nn.Module``): the attributes named in ``_TRANSIENT`` and everything called ``_tgmx_*`` are DERIVED state --
    ctypes blocks pointing into device buffers, workspaces, cached weight layouts -- and stay behind when the module is pickled or
    deep-copied (``copy.deepcopy(model)`` for a best-checkpoint copy or an EMA twin; ``torch.save(model)``): ctypes structures with
    pointers cannot be pickled at all, and a copy must not share the original's buffers.  The copy rebuilds them on its first call."""

    _TRANSIENT: Tuple[str, ...] = ()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)  # may REPLACE Parameter objects without any registration hook running
        global _registrations
        _registrations += 1
        return out

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in list(state):
            if k.startswith('_tgmx_'):
                del state[k]
            elif k in self._TRANSIENT:
                state[k] = None
        return state

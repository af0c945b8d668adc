from .base import BaseDGHook, DGHook, SeedableHook, StatefulHook, StatelessHook
from .dedup import DeduplicationHook
from .edge_list import SampledEdgeListHook
from .hook_manager import HookManager
from .negatives import RandomNegativeEdgeSamplerHook
from .recency import RecencyNeighborHook
from .uniform import NeighborSamplerHook
from .registry import hook, list_hooks
from . import neighbors  # noqa: E402,F401  (the reference's import path: tgm.hooks.neighbors.recency)

__all__ = [
    'BaseDGHook',
    'DGHook',
    'DeduplicationHook',
    'HookManager',
    'NeighborSamplerHook',
    'RandomNegativeEdgeSamplerHook',
    'RecencyNeighborHook',
    'SampledEdgeListHook',
    'SeedableHook',
    'StatefulHook',
    'StatelessHook',
    'hook',
    'list_hooks',
]

"""Import paths of the reference's sampler package (tgm/hooks/neighbors/__init__.py:1-2): ``tgm.hooks.neighbors.recency`` /
``.uniform`` resolve to the modules that hold the two samplers."""
import sys

from .. import recency, uniform
from ..recency import RecencyNeighborHook
from ..uniform import NeighborSamplerHook

for _m in (recency, uniform):
    sys.modules[f'{__name__}.{_m.__name__.rsplit(".", 1)[1]}'] = _m

__all__ = ['NeighborSamplerHook', 'RecencyNeighborHook']

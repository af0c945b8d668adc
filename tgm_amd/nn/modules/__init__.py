"""Import paths of the reference's module package (tgm/nn/modules/__init__.py:1-2) for the two modules on the hot path."""
import sys

from .. import attention, time_encoding
from ..attention import TemporalAttention
from ..time_encoding import Time2Vec

for _m in (attention, time_encoding):
    sys.modules[f'{__name__}.{_m.__name__.rsplit(".", 1)[1]}'] = _m

__all__ = ['TemporalAttention', 'Time2Vec']

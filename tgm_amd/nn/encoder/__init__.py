"""Import paths of the reference's encoder package (tgm/nn/encoder/__init__.py:1-15) for the encoders on the hot path:
``from tgm.nn.encoder.tgn import GraphAttentionEmbedding, LastAggregator, ...`` is how examples/linkproppred/tgn.py:25-31 reaches
them.  The submodule names resolve to the modules that hold the implementations -- nothing is defined here."""
import sys

from .. import tgat, tgcn, tgn
from ..tgat import TGAT
from ..tgcn import TGCN
from ..tgn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, MeanAggregator, TGNMemory

for _m in (tgat, tgcn, tgn):
    sys.modules[f'{__name__}.{_m.__name__.rsplit(".", 1)[1]}'] = _m

__all__ = ['GraphAttentionEmbedding', 'IdentityMessage', 'LastAggregator', 'MeanAggregator', 'TGAT', 'TGCN', 'TGNMemory']

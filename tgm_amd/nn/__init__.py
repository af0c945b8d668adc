from ._paramver import invalidate_parameter_caches
from .attention import TemporalAttention
from .base import EncoderModule
from .tgat import TGAT, MergeLayer
from .tgcn import TGCN, GCNConv
from .tgn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, MeanAggregator, TGNMemory, TGNStep, TransformerConv, sampled_edge_list
from .time_encoding import Time2Vec
from . import encoder, modules  # noqa: E402,F401  (the reference's import paths: tgm.nn.encoder.tgn, tgm.nn.modules.attention, ...)

__all__ = [
    'EncoderModule', 'GCNConv', 'GraphAttentionEmbedding', 'IdentityMessage', 'LastAggregator', 'MeanAggregator', 'MergeLayer', 'TGAT', 'TGCN',
    'TGNMemory', 'TGNStep', 'TemporalAttention', 'Time2Vec', 'TransformerConv', 'invalidate_parameter_caches', 'sampled_edge_list',
]  # fmt: skip

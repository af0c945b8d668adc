from .attention import TemporalAttention
from .base import EncoderModule
from .tgat import TGAT, MergeLayer
from .tgn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, MeanAggregator, TGNMemory, TransformerConv
from .time_encoding import Time2Vec

__all__ = [
    'EncoderModule', 'GraphAttentionEmbedding', 'IdentityMessage', 'LastAggregator', 'MeanAggregator', 'MergeLayer', 'TGAT',
    'TGNMemory', 'TemporalAttention', 'Time2Vec', 'TransformerConv',
]  # fmt: skip

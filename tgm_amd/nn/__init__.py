from .attention import TemporalAttention
from .base import EncoderModule
from .tgat import TGAT, MergeLayer
from .time_encoding import Time2Vec

__all__ = ['EncoderModule', 'MergeLayer', 'TGAT', 'TemporalAttention', 'Time2Vec']

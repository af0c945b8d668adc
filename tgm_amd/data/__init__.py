from .dg_data import DGData
from .loader import DGDataLoader
from .split import SplitStrategy, TemporalRatioSplit, TemporalSplit, TGBSplit

__all__ = ['DGData', 'DGDataLoader', 'SplitStrategy', 'TemporalRatioSplit', 'TemporalSplit', 'TGBSplit']

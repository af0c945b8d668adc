from .dg_data import DGData
from .loader import DGDataLoader

__all__ = ['DGData', 'DGDataLoader']

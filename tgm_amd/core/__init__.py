from .batch import DGBatch
from .graph import DGraph
from .timedelta import TimeDeltaDG

__all__ = ['DGBatch', 'DGraph', 'TimeDeltaDG']

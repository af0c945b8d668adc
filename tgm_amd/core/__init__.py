from .batch import DGBatch
from .graph import DGraph
from .lazy import EdgeFeaturesById
from .timedelta import TimeDeltaDG

__all__ = ['DGBatch', 'DGraph', 'EdgeFeaturesById', 'TimeDeltaDG']

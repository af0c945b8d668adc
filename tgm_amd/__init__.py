"""tgm_amd -- MI355X-native temporal neighbor sampling + TGAT/TGN aggregation.

Drop-in for the one hot path of tgm-team/tgm named in BASELINE.json: the
``RecencyNeighborHook`` neighbor sampler over a time-sorted COO edge store and
the TGAT / TGN aggregation that consumes it, behind the reference's own
``HookManager`` / ``DGBatch`` plugin surface.  All compute runs in hand-written
HIP kernels (gfx950) reached through the C ABI declared in ``include/tgm_amd.h``;
there is no CPU fallback: without ``libtgm_amd.so`` the hooks raise.
"""
from .constants import PADDED_NODE_ID
from .core.batch import DGBatch
from .core.graph import DGraph
from .core.timedelta import TimeDeltaDG
from .data.dg_data import DGData
from .data.loader import DGDataLoader

__all__ = ['PADDED_NODE_ID', 'DGBatch', 'DGraph', 'TimeDeltaDG', 'DGData', 'DGDataLoader']
__version__ = '0.1.0'

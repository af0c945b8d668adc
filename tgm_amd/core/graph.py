"""``DGraph`` -- an immutable (store, slice, device) view; API of tgm/core/graph.py:20-420.

The view itself is three Python objects; all tensors it hands out are
zero-copy windows of the device-resident :class:`EdgeStore` arrays.
"""
from __future__ import annotations

from functools import cached_property
from typing import Optional

import torch
from torch import Tensor

from .batch import DGBatch
from .store import DeviceArrays, EdgeStore, SliceBounds
from .timedelta import TimeDeltaDG


def _opt_max(a, b):
    return b if a is None else a if b is None else max(a, b)


def _opt_min(a, b):
    return b if a is None else a if b is None else min(a, b)


class DGraph:
    def __init__(self, data, device: str | torch.device = 'cpu') -> None:
        from ..data.dg_data import DGData  # circular import guard

        if not isinstance(data, DGData):
            raise TypeError(f'DGraph must be initialized with DGData, got {type(data)}')
        self._time_delta = data.time_delta
        self._storage = EdgeStore(data)
        self._device = torch.device(device)
        self._slice = SliceBounds()

    @classmethod
    def _view(cls, storage: EdgeStore, time_delta: TimeDeltaDG, device: torch.device, s: SliceBounds) -> 'DGraph':
        g = cls.__new__(cls)
        g._storage, g._time_delta, g._device, g._slice = storage, time_delta, device, s
        return g

    # -- views ------------------------------------------------------------
    def slice_events(self, start_idx: Optional[int] = None, end_idx: Optional[int] = None) -> 'DGraph':
        """New view over global event indices [start_idx, end_idx)."""
        if start_idx is not None and end_idx is not None and start_idx > end_idx:
            raise ValueError(f'start_idx ({start_idx}) must be <= end_idx ({end_idx})')
        s = self._slice.copy()
        s.start_idx = _opt_max(start_idx, s.start_idx)
        s.end_idx = _opt_min(end_idx, s.end_idx)
        return DGraph._view(self._storage, self._time_delta, self._device, s)

    def slice_time(self, start_time: Optional[int] = None, end_time: Optional[int] = None) -> 'DGraph':
        """New view over timestamps [start_time, end_time)."""
        if start_time is not None and end_time is not None and start_time > end_time:
            raise ValueError(f'start_time ({start_time}) must be <= end_time ({end_time})')
        if end_time is not None:
            end_time -= 1  # stored inclusive
        s = self._slice.copy()
        s.start_time = _opt_max(start_time, s.start_time)
        s.end_time = _opt_min(end_time, s.end_time)
        return DGraph._view(self._storage, self._time_delta, self._device, s)

    def to(self, device: str | torch.device) -> 'DGraph':
        return DGraph._view(self._storage, self._time_delta, torch.device(device), self._slice.copy())

    # -- scalar properties --------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def time_delta(self) -> TimeDeltaDG:
        return self._time_delta  # type: ignore[return-value]

    @cached_property
    def _event_range(self):
        lb, ub = self._storage.event_range(self._slice)
        return lb, max(lb, ub)

    @cached_property
    def _edge_range(self):
        lo, hi = self._storage.edge_range(self._slice)
        return lo, max(lo, hi)

    # (the reference writes the derived bound back into its slice tracker, graph.py:160-176; here the slice keeps only the
    # bounds the caller asked for -- the derived ones are implied by the index range, and a slice without time bounds is
    # what lets the loader address batches as plain edge ranges)
    @cached_property
    def start_time(self) -> Optional[int]:
        if self._slice.start_time is not None:
            return self._slice.start_time
        lb, ub = self._event_range
        return None if lb >= ub else self._storage.time_at(lb)

    @cached_property
    def end_time(self) -> Optional[int]:
        if self._slice.end_time is not None:
            return self._slice.end_time
        lb, ub = self._event_range
        return None if lb >= ub else self._storage.time_at(ub - 1)

    @cached_property
    def num_events(self) -> int:
        lb, ub = self._event_range
        return ub - lb

    @cached_property
    def num_edge_events(self) -> int:
        lo, hi = self._edge_range
        return hi - lo

    @cached_property
    def num_node_events(self) -> int:
        lo, hi = self._storage.node_x_range(self._slice)
        return max(0, hi - lo)

    @cached_property
    def num_node_labels(self) -> int:
        lo, hi = self._storage.node_y_range(self._slice)
        return max(0, hi - lo)

    @cached_property
    def num_timestamps(self) -> int:
        return self._storage.num_timestamps(self._slice)

    def __len__(self) -> int:
        return self.num_timestamps

    @cached_property
    def num_nodes(self) -> int:
        """max node id in the slice + 1 (0 for an empty slice)."""
        best = -1
        for t in (self.edge_src, self.edge_dst, self.node_x_nids):
            if t is not None and t.numel():
                best = max(best, int(t.max()))
        return best + 1

    @property
    def edge_x_dim(self) -> Optional[int]:
        return self._storage.edge_x_dim

    @property
    def static_node_x_dim(self) -> Optional[int]:
        return self._storage.static_node_x_dim

    @property
    def node_x_dim(self) -> Optional[int]:
        return self._storage.node_x_dim

    @property
    def node_y_dim(self) -> Optional[int]:
        return self._storage.node_y_dim

    # -- tensors (zero-copy windows of the resident arrays) ------------------
    @property
    def _arrays(self) -> DeviceArrays:
        return self._storage.on(self._device)

    def _edge_window(self, t: Optional[Tensor]) -> Optional[Tensor]:
        if t is None:
            return None
        lo, hi = self._edge_range
        return t.narrow(0, lo, hi - lo)

    @property
    def edge_src(self) -> Tensor:
        return self._edge_window(self._arrays.src)

    @property
    def edge_dst(self) -> Tensor:
        return self._edge_window(self._arrays.dst)

    @property
    def edge_time(self) -> Tensor:
        return self._edge_window(self._arrays.ts)

    @property
    def edge_x(self) -> Optional[Tensor]:
        lo, hi = self._edge_range
        if hi <= lo:
            return None
        return self._edge_window(self._arrays.edge_x)

    @property
    def edge_type(self) -> Optional[Tensor]:
        lo, hi = self._edge_range
        if hi <= lo:
            return None
        return self._edge_window(self._arrays.edge_type)

    def _node_window(self, t: Optional[Tensor], kind: str) -> Optional[Tensor]:
        if t is None:
            return None
        lo, hi = getattr(self._storage, f'node_{kind}_range')(self._slice)
        return t.narrow(0, lo, max(0, hi - lo))

    @property
    def node_x_nids(self) -> Optional[Tensor]:
        t = self._node_window(self._arrays.node_x_nids, 'x')
        return torch.empty(0, dtype=torch.int32, device=self._device) if t is None else t

    @property
    def node_x_time(self) -> Optional[Tensor]:
        t = self._node_window(self._arrays.node_x_time, 'x')
        return torch.empty(0, dtype=torch.int64, device=self._device) if t is None else t

    @property
    def node_x(self) -> Optional[Tensor]:
        t = self._node_window(self._arrays.node_x, 'x')
        return None if t is None or t.shape[0] == 0 else t

    @property
    def node_y_nids(self) -> Optional[Tensor]:
        t = self._node_window(self._arrays.node_y_nids, 'y')
        return torch.empty(0, dtype=torch.int32, device=self._device) if t is None else t

    @property
    def node_y_time(self) -> Optional[Tensor]:
        t = self._node_window(self._arrays.node_y_time, 'y')
        return torch.empty(0, dtype=torch.int64, device=self._device) if t is None else t

    @property
    def node_y(self) -> Optional[Tensor]:
        t = self._node_window(self._arrays.node_y, 'y')
        return None if t is None or t.shape[0] == 0 else t

    @property
    def static_node_x(self) -> Optional[Tensor]:
        return self._arrays.static_node_x

    @property
    def node_type(self) -> Optional[Tensor]:
        return self._arrays.node_type

    # -- batch --------------------------------------------------------------
    def materialize(self, materialize_features: bool = True) -> DGBatch:
        """Pack the slice into a ``DGBatch`` (tgm/core/graph.py:74-108).

        Dynamic node features are handed out dense (``node_x[i]`` belongs to
        event ``(node_x_time[i], node_x_nids[i])``), which is what the reference
        extracts from its sparse tensor via ``_indices()/_values()``.
        """
        arr = self._storage.on(self._device)
        lb, ub = self._storage.event_range(self._slice)
        if self._storage.num_edges == self._storage.num_events:
            lo, hi = lb, max(lb, ub)
        else:
            lo, hi = self._edge_range
        n = hi - lo
        batch = DGBatch(arr.src.narrow(0, lo, n), arr.dst.narrow(0, lo, n), arr.ts.narrow(0, lo, n))
        batch._edge_lo = lo
        batch._event_lo = lb
        if materialize_features and arr.node_x is not None and self.node_x is not None:
            batch.node_x_time, batch.node_x_nids, batch.node_x = self.node_x_time, self.node_x_nids, self.node_x
        if materialize_features and n > 0 and arr.edge_x is not None:
            batch.edge_x = arr.edge_x.narrow(0, lo, n)
        if materialize_features and arr.node_y is not None and self.node_y is not None:
            batch.node_y_time, batch.node_y_nids, batch.node_y = self.node_y_time, self.node_y_nids, self.node_y
        if n > 0 and arr.edge_type is not None:
            batch.edge_type = arr.edge_type.narrow(0, lo, n)
        return batch

    def __str__(self) -> str:
        return f'DGraph(storage=EdgeStore, time_delta={self.time_delta}, device={self.device})'

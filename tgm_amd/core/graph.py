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
    def _from_storage(cls, storage: EdgeStore, time_delta: TimeDeltaDG, device: torch.device, slice: SliceBounds) -> 'DGraph':
        """A view over an existing store (tgm/core/graph.py:406-420; the store is shared, never copied)."""
        g = cls.__new__(cls)
        g._storage, g._time_delta, g._device, g._slice = storage, time_delta, device, slice
        return g

    # -- views ------------------------------------------------------------
    def slice_events(self, start_idx: Optional[int] = None, end_idx: Optional[int] = None) -> 'DGraph':
        """New view over global event indices [start_idx, end_idx)."""
        if start_idx is not None and end_idx is not None and start_idx > end_idx:
            raise ValueError(f'start_idx ({start_idx}) must be <= end_idx ({end_idx})')
        s = self._slice.copy()
        s.start_idx = _opt_max(start_idx, s.start_idx)
        s.end_idx = _opt_min(end_idx, s.end_idx)
        return DGraph._from_storage(self._storage, self._time_delta, self._device, s)

    def slice_time(self, start_time: Optional[int] = None, end_time: Optional[int] = None) -> 'DGraph':
        """New view over timestamps [start_time, end_time)."""
        if start_time is not None and end_time is not None and start_time > end_time:
            raise ValueError(f'start_time ({start_time}) must be <= end_time ({end_time})')
        if end_time is not None:
            end_time -= 1  # stored inclusive
        s = self._slice.copy()
        s.start_time = _opt_max(start_time, s.start_time)
        s.end_time = _opt_min(end_time, s.end_time)
        return DGraph._from_storage(self._storage, self._time_delta, self._device, s)

    def to(self, device: str | torch.device) -> 'DGraph':
        return DGraph._from_storage(self._storage, self._time_delta, torch.device(device), self._slice.copy())

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
    # A view is immutable and bound to one device, so every window is cut once and cached: like the reference's cached
    # ``_edges_cpu`` + ``.to(same device)`` (graph.py:228-248), two reads of ``dg.edge_src`` -- and two ``materialize()`` calls --
    # hand out the SAME tensor object, which is what ``batch == batch`` (dataclass ``==`` over tensors) relies on.
    @property
    def _arrays(self) -> DeviceArrays:
        return self._storage.on(self._device)

    @cached_property
    def _edges(self):
        arr = self._arrays
        lo, hi = self._edge_range
        n = hi - lo
        return arr.src.narrow(0, lo, n), arr.dst.narrow(0, lo, n), arr.ts.narrow(0, lo, n)

    @property
    def edge_src(self) -> Tensor:
        return self._edges[0]

    @property
    def edge_dst(self) -> Tensor:
        return self._edges[1]

    @property
    def edge_time(self) -> Tensor:
        return self._edges[2]

    def _edge_window(self, t: Optional[Tensor]) -> Optional[Tensor]:
        lo, hi = self._edge_range
        if t is None or hi <= lo:  # the reference hands out None for a slice without edges (array_backend.py:262-285)
            return None
        return t.narrow(0, lo, hi - lo)

    @cached_property
    def edge_x(self) -> Optional[Tensor]:
        return self._edge_window(self._arrays.edge_x)

    @cached_property
    def edge_type(self) -> Optional[Tensor]:
        return self._edge_window(self._arrays.edge_type)

    @cached_property
    def _node_events(self):
        """(nids, time, rows) windows of the dynamic node features in the slice; empty id / time tensors when there are none."""
        return self._event_windows('x')

    @cached_property
    def _node_labels(self):
        return self._event_windows('y')

    def _event_windows(self, kind: str):
        arr = self._arrays
        nids, time, rows = getattr(arr, f'node_{kind}_nids'), getattr(arr, f'node_{kind}_time'), getattr(arr, f'node_{kind}')
        if nids is None:
            return torch.empty(0, dtype=torch.int32, device=self._device), torch.empty(0, dtype=torch.int64, device=self._device), None
        lo, hi = getattr(self._storage, f'node_{kind}_range')(self._slice)
        n = max(0, hi - lo)
        return nids.narrow(0, lo, n), time.narrow(0, lo, n), (rows.narrow(0, lo, n) if n else None)

    @property
    def node_x_nids(self) -> Tensor:
        return self._node_events[0]

    @property
    def node_x_time(self) -> Tensor:
        return self._node_events[1]

    @property
    def node_y_nids(self) -> Tensor:
        return self._node_labels[0]

    @property
    def node_y_time(self) -> Tensor:
        return self._node_labels[1]

    def _sparse_events(self, kind: str) -> Optional[Tensor]:
        """``sparse_coo_tensor(T x V x d)`` over the slice's node events, built from the resident windows (no copy of the rows):
        indices ``[time ; node]``, T = the slice's end time + 1, V = the largest node id among the slice's edges, node events and
        node labels + 1 (tgm/core/_storage/backends/array_backend.py:178-256)."""
        nids, time, rows = self._node_events if kind == 'x' else self._node_labels
        if rows is None:
            return None
        tops = [nids.max()]
        src, dst, _ = self._edges
        if src.numel():
            tops += [src.max(), dst.max()]
        other = (self._node_labels if kind == 'x' else self._node_events)[0]
        if other.numel():
            tops.append(other.max())
        max_node = int(torch.stack(tops).max())
        _, ub = self._event_range
        max_time = self._slice.end_time or self._storage.time_at(ub - 1)
        return torch.sparse_coo_tensor(torch.stack([time, nids.to(torch.int64)]), rows, (max_time + 1, max_node + 1, rows.shape[1]))

    @cached_property
    def node_x(self) -> Optional[Tensor]:
        """Dynamic node features as the reference's ``sparse_coo_tensor(T x V x d_node_dynamic)`` (graph.py:300-309)."""
        return self._sparse_events('x')

    @cached_property
    def node_y(self) -> Optional[Tensor]:
        """Dynamic node labels as the reference's ``sparse_coo_tensor(T x V x d_node_label)`` (graph.py:347-356)."""
        return self._sparse_events('y')

    @property
    def static_node_x(self) -> Optional[Tensor]:
        return self._arrays.static_node_x

    @property
    def node_type(self) -> Optional[Tensor]:
        return self._arrays.node_type

    # -- batch --------------------------------------------------------------
    def materialize(self, materialize_features: bool = True) -> DGBatch:
        """Pack the slice into a ``DGBatch`` (tgm/core/graph.py:74-108).

        The batch carries the view's cached windows.  ``batch.node_x`` is the dense ``[n, d]`` row window and
        ``node_x_time / node_x_nids`` its coordinates -- exactly the ``_values()`` / ``_indices()`` of ``dg.node_x`` the
        reference unpacks, without building the sparse tensor on the per-batch path.
        """
        src, dst, ts = self._edges
        batch = DGBatch(src, dst, ts)
        batch._edge_lo = self._edge_range[0]
        batch._event_lo = self._event_range[0]
        if materialize_features:
            nids, time, rows = self._node_events
            if rows is not None:
                batch.node_x_time, batch.node_x_nids, batch.node_x = time, nids, rows
            batch.edge_x = self.edge_x
            nids, time, rows = self._node_labels
            if rows is not None:
                batch.node_y_time, batch.node_y_nids, batch.node_y = time, nids, rows
        batch.edge_type = self.edge_type
        return batch

    def __str__(self) -> str:
        return f'DGraph(storage=EdgeStore, time_delta={self.time_delta}, device={self.device})'

"""Chronological train / val / test splits of a ``DGData`` (role of tgm/data/split.py:99-294; every example script starts with
``train, val, test = data.split()``).

The timeline of a ``DGData`` is sorted, so a split interval ``[start, end)`` is ONE contiguous range of every event group -- edges,
dynamic node features, node labels -- found by a binary search per group; a split is built from ``narrow`` views of the parent's
arrays and handed to ``DGData.from_raw`` (which re-derives the merged timeline).  Static node features and node types are shared,
not copied.  Semantics kept from the reference: train = times below ``val_time``, val = ``[val_time, test_time)``, test = the rest;
a split without any edge is dropped from the returned tuple; a split left without node events / labels carries ``None`` for them;
ratio splits cut the time SPAN (not the event count) and hand the two boundaries to ``TemporalSplit``; ``TGBSplit`` takes closed edge-time
intervals per split and gives node labels the interval ``[start - 1, end)``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor


def _group_times(data, pos: Optional[Tensor]) -> Optional[Tensor]:
    """timestamps of one event group (its positions in the merged timeline are ascending, so these are sorted)"""
    return None if pos is None else data.time[pos.long()]


def _range(times: Optional[Tensor], start: float, end: float) -> Optional[Tuple[int, int]]:
    """[lo, hi) of the sorted ``times`` with start <= t < end (infinite bounds allowed)"""
    if times is None:
        return None
    n = int(times.shape[0])
    lo = 0 if math.isinf(start) and start < 0 else int(torch.searchsorted(times, torch.tensor(int(start), dtype=times.dtype)))
    hi = n if math.isinf(end) and end > 0 else int(torch.searchsorted(times, torch.tensor(int(end), dtype=times.dtype)))
    return lo, max(lo, hi)


class SplitStrategy:
    """A rule that cuts a ``DGData`` into chronological parts (``apply`` returns them as a tuple)."""

    def apply(self, data) -> Tuple['DGData', ...]:  # noqa: F821
        raise NotImplementedError

    @staticmethod
    def _part(data, edges: Tuple[int, int], node_x: Optional[Tuple[int, int]], node_y: Optional[Tuple[int, int]]):
        """The events of ``data`` in the given per-group ranges as a new ``DGData`` (None for a node group: all of it)."""
        from .dg_data import DGData

        cut = lambda t, r: None if t is None else t.narrow(0, r[0], r[1] - r[0])
        kw = dict(edge_time=cut(_group_times(data, data.edge_mask), edges), edge_index=cut(data.edge_index, edges), edge_x=cut(data.edge_x, edges),
                  edge_type=cut(data.edge_type, edges), static_node_x=data.static_node_x, node_type=data.node_type, time_delta=data.time_delta)
        for kind, r in (('node_x', node_x), ('node_y', node_y)):
            nids = getattr(data, f'{kind}_nids')
            if nids is None:
                continue
            if r is None:
                r = (0, int(nids.shape[0]))
            if r[1] > r[0]:  # (a split without events of this kind carries None for the whole group)
                kw[f'{kind}_time'] = cut(_group_times(data, getattr(data, f'{kind}_mask')), r)
                kw[f'{kind}_nids'] = cut(nids, r)
                kw[kind] = cut(getattr(data, kind), r)
        return DGData.from_raw(**kw)


@dataclass
class TemporalSplit(SplitStrategy):
    """Absolute boundaries: train = t < val_time, val = val_time <= t < test_time, test = t >= test_time."""

    val_time: int
    test_time: int

    def __post_init__(self) -> None:
        if not 0 <= self.val_time <= self.test_time:
            raise ValueError(f'Expected 0 <= val_time <= test_time, got {self.val_time}, {self.test_time}')

    def apply(self, data) -> Tuple['DGData', ...]:  # noqa: F821
        groups = [_group_times(data, m) for m in (data.edge_mask, data.node_x_mask if data.node_x_nids is not None else None,
                                                  data.node_y_mask if data.node_y_nids is not None else None)]
        parts = []
        for start, end in ((-math.inf, self.val_time), (self.val_time, self.test_time), (self.test_time, math.inf)):
            edges = _range(groups[0], start, end)
            if edges[1] == edges[0]:
                continue  # no edge in this interval: the split does not exist
            parts.append(self._part(data, edges, _range(groups[1], start, end), _range(groups[2], start, end)))
        return tuple(parts)


@dataclass
class TemporalRatioSplit(SplitStrategy):
    """Fractions of the time SPAN ``[time[0], time[-1]]``, applied cumulatively; the boundaries go through ``TemporalSplit``."""

    train_ratio: float = 0.7
    val_ratio: float = 0.15
    test_ratio: float = 0.15

    def __post_init__(self) -> None:
        if min(self.train_ratio, self.val_ratio, self.test_ratio) < 0:
            raise ValueError('Ratios must all be non-negative')
        total = self.train_ratio + self.val_ratio + self.test_ratio
        if abs(total - 1.0) > 1e-6:
            raise ValueError(f'train_ratio + val_ratio + test_ratio must sum to 1.0, got {total}')

    def apply(self, data) -> Tuple['DGData', ...]:  # noqa: F821
        first, last = data.time[0], data.time[-1]
        span = last - first + 1
        val_time = first + int(span * self.train_ratio)
        return TemporalSplit(val_time=val_time, test_time=val_time + int(span * self.val_ratio)).apply(data)


@dataclass
class TGBSplit(SplitStrategy):
    """Closed edge-time intervals per split name (what a TGB dataset ships); always three parts, empty ones included."""

    split_bounds: Dict[str, Tuple[int, int]]

    def apply(self, data) -> Tuple['DGData', ...]:  # noqa: F821
        edge_t = _group_times(data, data.edge_mask)
        label_t = _group_times(data, data.node_y_mask) if data.node_y_nids is not None else None
        parts = []
        for name in ('train', 'val', 'test'):
            lo_t, hi_t = self.split_bounds[name]
            edges = _range(edge_t, lo_t, hi_t + 1)
            labels = _range(label_t, lo_t - 1, hi_t) if (label_t is not None and edges[1] > edges[0]) else None
            parts.append(self._part(data, edges, None, labels))
        return tuple(parts)

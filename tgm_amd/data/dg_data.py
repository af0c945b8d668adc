"""``DGData`` -- host-side ingest container for a temporal graph.

Keeps the reference's field names and normalisation rules
(tgm/data/dg_data.py:29-84 fields, :86-394 validation) because the rest of
the drop-in surface (``DGraph``, hooks) reads these fields: one global,
sorted ``time[int64]`` timeline over all events; ``edge_mask[int32]`` =
position of every edge in that timeline; ``edge_index[int32, E x 2]``;
``edge_x[float32, E x D]``; optional dynamic node events / labels and static
node features.  Ingest is one-shot host work and is deliberately plain torch;
everything per-batch lives on the device (see ``core/store.py``).

Only ``from_raw`` is provided (csv / pandas / TGB loaders are out of scope,
SURVEY.md section 2.1); ``split()`` cuts the sorted timeline into contiguous
ranges (``data/split.py``).
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor

from ..constants import PADDED_NODE_ID
from ..core.timedelta import TimeDeltaDG
from ..exceptions import EmptyGraphError, InvalidNodeIDError
from .split import TemporalRatioSplit, TGBSplit  # (split.py reaches back for DGData inside its functions only)

_INT_DTYPES = (torch.int8, torch.int16, torch.int32, torch.int64, torch.uint8)
_I32_MAX = torch.iinfo(torch.int32).max


def _need_tensor(x, name: str) -> None:
    if not isinstance(x, Tensor):
        raise TypeError(f'{name} must be a Tensor, got: {type(x)}')
    if x.is_floating_point() and torch.isnan(x).any():
        raise ValueError(f'{name} contains NaN values')


def _need_integral(x: Tensor, name: str) -> None:
    if x.dtype not in _INT_DTYPES:
        raise TypeError(f'{name} must have integer dtype but got: {x.dtype}')


def _as_f32(x: Tensor, name: str) -> Tensor:
    if x.dtype == torch.float64:
        warnings.warn(f'Downcasting {name} from torch.float64 to torch.float32', UserWarning)
    return x if x.dtype == torch.float32 else x.to(torch.float32)


def _as_i32(x: Tensor, name: str) -> Tensor:
    if x.dtype == torch.int64:
        warnings.warn(f'Downcasting {name} from torch.int64 to torch.int32', UserWarning)
    return x if x.dtype == torch.int32 else x.to(torch.int32)


def _check_node_ids(x: Tensor, what: str) -> None:
    if torch.any(x == PADDED_NODE_ID):
        raise InvalidNodeIDError(
            f'{what} contains node ids matching PADDED_NODE_ID: {PADDED_NODE_ID}, which marks empty '
            'neighbor slots. Remap node ids to non-negative integers.'
        )
    if not torch.all(x < _I32_MAX):
        raise InvalidNodeIDError(f'{what} contains node ids that exceed the int32 limit ({_I32_MAX}).')


@dataclass
class DGData:
    time_delta: TimeDeltaDG | str
    time: Tensor  # [num_events] int64, sorted

    edge_mask: Tensor  # [E] int32 -> index into ``time``
    edge_index: Tensor  # [E, 2] int32
    edge_x: Optional[Tensor] = None  # [E, D] float32

    node_x_mask: Optional[Tensor] = None
    node_x_nids: Optional[Tensor] = None
    node_x: Optional[Tensor] = None

    node_y_mask: Optional[Tensor] = None
    node_y_nids: Optional[Tensor] = None
    node_y: Optional[Tensor] = None

    static_node_x: Optional[Tensor] = None  # [N, d0] float32
    edge_type: Optional[Tensor] = None
    node_type: Optional[Tensor] = None

    _split_strategy: Optional[object] = None  # a data set that ships its own split (TGB) pins it here (dg_data.py:54)

    def __post_init__(self) -> None:
        if isinstance(self.time_delta, str):
            self.time_delta = TimeDeltaDG(self.time_delta)

        _need_tensor(self.time, 'timestamps')
        _need_integral(self.time, 'timestamps')
        if not torch.all(self.time >= 0):
            raise ValueError('timestamps must all be non-negative')
        if not torch.all(self.time < _I32_MAX):
            raise ValueError(f'timestamps exceed the int32 limit ({_I32_MAX}).')
        self.time = self.time.to(torch.int64)
        if len(self.time) > _I32_MAX:
            raise ValueError(f'Number of events ({len(self.time)}) exceeds the int32 limit ({_I32_MAX}).')

        _need_tensor(self.edge_index, 'edge_index')
        _need_integral(self.edge_index, 'edge_index')
        if self.edge_index.ndim != 2 or self.edge_index.shape[1] != 2:
            raise ValueError(f'edge_index must have shape [num_edges, 2], got: {self.edge_index.shape}')
        _check_node_ids(self.edge_index, 'Edge events')
        self.edge_index = _as_i32(self.edge_index, 'edge_index')
        E = self.edge_index.shape[0]
        if E == 0:
            raise EmptyGraphError('graphs without edge events are not supported')

        _need_tensor(self.edge_mask, 'edge_mask')
        _need_integral(self.edge_mask, 'edge_mask')
        self.edge_mask = self.edge_mask.to(torch.int32)

        if self.edge_x is not None:
            _need_tensor(self.edge_x, 'edge_x')
            if self.edge_x.ndim != 2 or self.edge_x.shape[0] != E:
                raise ValueError(f'edge features must have shape [num_edges, D_edge], got {E} edges and shape {self.edge_x.shape}')
            self.edge_x = _as_f32(self.edge_x, 'edge_x')

        n_node_events = self._check_node_events('node_x')
        n_node_labels = self._check_node_events('node_y')

        num_nodes = int(self.edge_index.max()) + 1
        if self.node_x_nids is not None:
            num_nodes = max(num_nodes, int(self.node_x_nids.max()) + 1)
        if self.node_y_nids is not None and int(self.node_y_nids.max()) + 1 > num_nodes:
            raise InvalidNodeIDError(
                f"Dynamic node labels reference node ids outside the graph's range (max label id "
                f'{int(self.node_y_nids.max())}, num_nodes {num_nodes}).'
            )

        if self.static_node_x is not None:
            _need_tensor(self.static_node_x, 'static_node_x')
            if self.static_node_x.ndim != 2:
                raise ValueError(f'static_node_x must be [N, D_node_static], got shape {self.static_node_x.shape}')
            if self.static_node_x.shape[0] < num_nodes:
                raise ValueError(
                    f'static_node_x has shape {self.static_node_x.shape} but the data needs features for at least {num_nodes} nodes'
                )
            self.static_node_x = _as_f32(self.static_node_x, 'static_node_x')

        if self.edge_type is not None:
            _need_tensor(self.edge_type, 'edge_type')
            _need_integral(self.edge_type, 'edge_type')
            if self.edge_type.ndim != 1 or self.edge_type.shape[0] != E:
                raise ValueError(f'edge_type must have shape [num_edges], got {self.edge_type.shape}')
        if self.node_type is not None:
            _need_tensor(self.node_type, 'node_type')
            _need_integral(self.node_type, 'node_type')
            if self.node_type.ndim != 1 or self.node_type.shape[0] < num_nodes:
                raise ValueError(f'node_type must have shape [num_nodes], got {self.node_type.shape}')

        if self.time.ndim != 1 or self.time.shape[0] != E + n_node_events + n_node_labels:
            raise ValueError(
                f'time must have shape [num_edges + num_node_events + num_node_labels], got {E} edges, '
                f'{n_node_events} node events, {n_node_labels} node labels, shape {self.time.shape}'
            )

        if not torch.all(self.time[1:] >= self.time[:-1]):
            self._sort_timeline()

    # ------------------------------------------------------------------
    def _check_node_events(self, prefix: str) -> int:
        mask = getattr(self, f'{prefix}_mask')
        if mask is None:
            return 0
        _need_tensor(mask, f'{prefix}_mask')
        _need_integral(mask, f'{prefix}_mask')
        mask = mask.to(torch.int32)
        setattr(self, f'{prefix}_mask', mask)
        n = mask.shape[0]
        if n == 0:
            raise ValueError(f'{prefix}_mask is an empty tensor, please double-check your inputs')
        nids = getattr(self, f'{prefix}_nids')
        _need_tensor(nids, f'{prefix}_nids')
        _need_integral(nids, f'{prefix}_nids')
        if nids.ndim != 1 or nids.shape[0] != n:
            raise ValueError(f'{prefix}_nids must have shape [{n}], got {nids.shape}')
        _check_node_ids(nids, f'{prefix} events')
        setattr(self, f'{prefix}_nids', _as_i32(nids, f'{prefix}_nids'))
        vals = getattr(self, prefix)
        if vals is not None:
            _need_tensor(vals, prefix)
            if vals.ndim != 2 or vals.shape[0] != n:
                raise ValueError(f'{prefix} must have shape [{n}, D], got {vals.shape}')
            setattr(self, prefix, _as_f32(vals, prefix))
        return n

    def _sort_timeline(self) -> None:
        """Reorder every event array to time-sorted order.  Events with EQUAL timestamps come out in the order ``torch.argsort``'s default
        (unstable) host sort leaves them in -- the reference's own call (tgm/data/dg_data.py:356), so unsorted input with ties is
        reordered exactly as the reference reorders it under the same torch build (pinned by fixture g13_dgraph_views); the
        per-group regrouping below sorts unique positions, where stability is moot."""
        warnings.warn('Timestamps in DGData are not globally sorted; reordering all events', UserWarning)
        order = torch.argsort(self.time)
        rank = torch.empty_like(order)
        rank[order] = torch.arange(len(order))
        self.time = self.time[order]

        def regroup(mask: Tensor, *payload):
            new_pos = rank[mask.long()]
            o = torch.argsort(new_pos, stable=True)
            return (new_pos[o].to(torch.int32),) + tuple(None if p is None else p[o] for p in payload)

        self.edge_mask, self.edge_index, self.edge_x, self.edge_type = regroup(
            self.edge_mask, self.edge_index, self.edge_x, self.edge_type
        )
        if self.node_x_mask is not None:
            self.node_x_mask, self.node_x_nids, self.node_x = regroup(self.node_x_mask, self.node_x_nids, self.node_x)
        if self.node_y_mask is not None:
            self.node_y_mask, self.node_y_nids, self.node_y = regroup(self.node_y_mask, self.node_y_nids, self.node_y)

    # ------------------------------------------------------------------
    def split(self, strategy=None):
        """Chronological parts of the data set (tgm/data/dg_data.py:396-421): ``strategy``, else the data set's own, else
        ``TemporalRatioSplit()`` (70 / 15 / 15 % of the time span).  A data set that ships its split cannot be split another way."""
        strategy = strategy or self._split_strategy or TemporalRatioSplit()
        if isinstance(self._split_strategy, TGBSplit) and strategy is not self._split_strategy:
            raise ValueError('Cannot override split strategy for TGB datasets')
        return strategy.apply(self)

    def clone(self) -> 'DGData':
        """Deep copy (tensors cloned)."""
        import copy
        from dataclasses import fields

        vals = {f.name: getattr(self, f.name) for f in fields(self)}
        return DGData(**{k: (v.clone() if isinstance(v, Tensor) else copy.deepcopy(v)) for k, v in vals.items()})

    def discretize(self, time_delta: 'TimeDeltaDG | str | None', reduce_op: str = 'first', device=None) -> 'DGData':
        """Coarsen the time granularity (tgm/data/dg_data.py:423-564): timestamps become bucket indices
        ``floor(t * self.time_delta / time_delta)`` and, per bucket, only the FIRST event of every
        (src, dst) edge / node id is kept, in chronological order.

        The grouping is sort-bound.  With ``device=`` a ROCm device (ours; or when the tensors already live on one) every
        event group goes through ``tgmx_discretize_keep`` (``csrc/discretize.hip``: the reference's int32 radix key, one
        stable rocPRIM radix sort, first-of-group marks compacted in event order); host tensors without ``device=`` keep
        the reference's torch formulation (host data plumbing, like the reference).  Both give identical results."""
        from ..exceptions import EventOrderedConversionError, InvalidDiscretizationError

        if isinstance(time_delta, str):
            time_delta = TimeDeltaDG(time_delta)
        if time_delta is None or self.time_delta == time_delta:
            return self.clone()
        if self.time_delta.is_event_ordered or time_delta.is_event_ordered:
            raise EventOrderedConversionError('Cannot discretize a graph with event-ordered time granularity')
        if self.time_delta.is_coarser_than(time_delta):
            raise InvalidDiscretizationError(f'Cannot discretize to {time_delta} which is strictly finer than {self.time_delta}')
        if reduce_op != 'first':
            raise ValueError(f"Unknown reduce_op: {reduce_op}, expected one of: ['first']")

        factor = self.time_delta.convert(time_delta)
        buckets = (self.time.to(torch.float64) * factor).floor().int()  # float64: exact for every int32 timestamp

        dev = torch.device(device) if device is not None else self.time.device
        if dev.type == 'cuda':
            return self._discretize_on_device(time_delta, factor, dev)

        def first_per_bucket(event_pos: Tensor, ids: Tensor) -> Tensor:
            """positions (within this event group) of the first event of every (bucket, id) pair, ascending"""
            b = buckets[event_pos.long()]
            if ids.ndim == 2:
                base = int(ids.max()) + 1
                flat = ids[:, 0] * base + ids[:, 1]  # int32 arithmetic, like the reference
            else:
                flat = ids
            key = b * (int(flat.max()) + 1) + flat
            skey, order = torch.sort(key, stable=True)
            head = torch.ones_like(skey, dtype=torch.bool)
            head[1:] = skey[1:] != skey[:-1]
            return order[head].sort().values

        keep = first_per_bucket(self.edge_mask, self.edge_index)
        pick = lambda t: None if t is None else t[keep]
        kw = dict(
            time_delta=time_delta, edge_time=buckets[self.edge_mask.long()][keep], edge_index=self.edge_index[keep],
            edge_x=pick(self.edge_x), edge_type=pick(self.edge_type),
            static_node_x=None if self.static_node_x is None else self.static_node_x.clone(),
            node_type=None if self.node_type is None else self.node_type.clone(),
        )  # fmt: skip
        for pre in ('node_x', 'node_y'):
            mask = getattr(self, f'{pre}_mask')
            if mask is not None:
                kp = first_per_bucket(mask, getattr(self, f'{pre}_nids'))
                kw[f'{pre}_time'] = buckets[mask.long()][kp]
                kw[f'{pre}_nids'] = getattr(self, f'{pre}_nids')[kp]
                vals = getattr(self, pre)
                kw[pre] = None if vals is None else vals[kp]
        return DGData.from_raw(**kw)

    def _discretize_on_device(self, time_delta: 'TimeDeltaDG', factor: float, dev: torch.device) -> 'DGData':
        """``discretize`` with every event group's grouping in ``tgmx_discretize_keep``; the result lives where ``self`` does."""
        from .. import _native

        lib = _native.load()
        home = self.time.device
        up = lambda t: t.to(dev).contiguous()
        time_d = up(self.time)

        def group(event_pos: Tensor, ids: Tensor):
            """(kept positions within the group, their buckets), both on the device"""
            n = int(event_pos.numel())
            t = time_d[up(event_pos).long()].contiguous()
            ids = up(ids).to(torch.int32)
            id0 = ids[:, 0].contiguous() if ids.ndim == 2 else ids
            id1 = ids[:, 1].contiguous() if ids.ndim == 2 else None
            if n and int(ids.min()) < 0:
                raise ValueError('discretize: ids must be >= 0')
            bucket = torch.empty(n, dtype=torch.int32, device=dev)
            keep = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
            count = torch.zeros(1, dtype=torch.int64, device=dev)
            need = int(lib.tgmx_discretize_workspace_bytes(n))
            if need == 0:
                _native.check(-2, 'tgmx_discretize_workspace_bytes')
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _native.check(lib.tgmx_discretize_keep(t.data_ptr(), id0.data_ptr(), _native.ptr(id1), n, float(factor), bucket.data_ptr(),
                                                       keep.data_ptr(), count.data_ptr(), ws.data_ptr(), need, _native.stream_ptr(dev.index)),
                              'tgmx_discretize_keep')  # fmt: skip
            keep = keep[: int(count.item())]
            return keep, bucket[keep]

        back = lambda t: t.to(home)
        keep, b_keep = group(self.edge_mask, self.edge_index)
        pick = lambda t, kp: None if t is None else back(up(t)[kp])
        kw = dict(
            time_delta=time_delta, edge_time=back(b_keep), edge_index=pick(self.edge_index, keep), edge_x=pick(self.edge_x, keep),
            edge_type=pick(self.edge_type, keep), static_node_x=None if self.static_node_x is None else self.static_node_x.clone(),
            node_type=None if self.node_type is None else self.node_type.clone(),
        )  # fmt: skip
        for pre in ('node_x', 'node_y'):
            mask = getattr(self, f'{pre}_mask')
            if mask is not None:
                kp, bk = group(mask, getattr(self, f'{pre}_nids'))
                kw[f'{pre}_time'] = back(bk)
                kw[f'{pre}_nids'] = pick(getattr(self, f'{pre}_nids'), kp)
                kw[pre] = pick(getattr(self, pre), kp)
        return DGData.from_raw(**kw)

    @property
    def num_nodes(self) -> int:
        n = int(self.edge_index.max()) + 1
        if self.node_x_nids is not None:
            n = max(n, int(self.node_x_nids.max()) + 1)
        return n

    @classmethod
    def from_raw(
        cls,
        edge_time: Tensor,
        edge_index: Tensor,
        edge_x: Optional[Tensor] = None,
        node_x_time: Optional[Tensor] = None,
        node_x_nids: Optional[Tensor] = None,
        node_x: Optional[Tensor] = None,
        node_y_time: Optional[Tensor] = None,
        node_y_nids: Optional[Tensor] = None,
        node_y: Optional[Tensor] = None,
        static_node_x: Optional[Tensor] = None,
        time_delta: TimeDeltaDG | str = 'r',
        edge_type: Optional[Tensor] = None,
        node_type: Optional[Tensor] = None,
    ) -> 'DGData':
        """Concatenate edge / node-event / node-label timelines (in that order) and
        record each group's positions; same argument list as tgm/data/dg_data.py:592-607."""
        for name, t in (('edge_time', edge_time), ('node_x_time', node_x_time), ('node_y_time', node_y_time)):
            if t is not None:
                _need_tensor(t, name)
        E = edge_time.shape[0]
        pieces = [edge_time]
        edge_mask = torch.arange(E)
        node_x_mask = node_y_mask = None
        off = E
        if node_x_time is not None:
            pieces.append(node_x_time)
            node_x_mask = torch.arange(off, off + node_x_time.shape[0])
            off += node_x_time.shape[0]
        if node_y_time is not None:
            pieces.append(node_y_time)
            node_y_mask = torch.arange(off, off + node_y_time.shape[0])
        time = torch.cat([p.to(torch.int64) if p.dtype in _INT_DTYPES else p for p in pieces])
        return cls(
            time_delta=time_delta,
            time=time,
            edge_mask=edge_mask,
            edge_index=edge_index,
            edge_x=edge_x,
            node_x_mask=node_x_mask,
            node_x_nids=node_x_nids,
            node_x=node_x,
            node_y_mask=node_y_mask,
            node_y_nids=node_y_nids,
            node_y=node_y,
            static_node_x=static_node_x,
            edge_type=edge_type,
            node_type=node_type,
        )

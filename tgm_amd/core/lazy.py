"""``EdgeFeaturesById`` -- the sampler's ``nbr_edge_x`` without the copy.

The reference gathers every sampled neighbor's edge-feature row into dense ``[S, k, D]`` tensors
(tgm/hooks/neighbors/recency.py:287-319); at the headline shape that copy is 177 of the 244 MB a batch moves, and
the attention then reads it once.  With ``RecencyNeighborHook(edge_features='by_id')`` the sampler publishes the EDGE ID
behind every slot instead (``tgmx_recency_step_t.out_eid``) and ``batch.nbr_edge_x`` is this object: a list whose items
are materialized (one gather from the resident store, cached) only when somebody indexes it.  ``tgm_amd.nn.TGAT``
does not: its attention kernel reads ``edge_x[eid]`` where it consumes the row (``tgmx_tgat_hop_t.nbr_eid``).
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor


class SamplerCallTag:
    """Identity of ONE sampler call, shared by the per-hop lists it published (``SampledHops``, ``EdgeFeaturesById``).

    ``tgm_amd.nn.TGAT`` uses it to recognise inputs whose deeper hops were sampled FOR the shallower hops' outputs by one call --
    then slots with the same (neighbor id, time) are the same row of every deeper level and inference computes each distinct row
    once (``tgmx_tgat_hop_t.seed_keyed``).  The stamp records every output tensor's (address, version, rows) -- ids, times and edge features (or, by id, the edge ids): a list whose items
    were replaced, or a tensor modified in place through torch, no longer matches and gets the row-per-slot computation."""

    __slots__ = ('stamp',)

    def __init__(self, nbr_nids: List[Tensor], nbr_edge_time: List[Tensor], nbr_edge_x=None) -> None:
        self.stamp = self._of(nbr_nids, nbr_edge_time, nbr_edge_x)

    @staticmethod
    def _of(nbr_nids, nbr_edge_time, nbr_edge_x=None) -> tuple:
        # edge features: the dense copies themselves, or (by id) the edge ids the rows are read through
        feats = () if nbr_edge_x is None else (nbr_edge_x.eids if hasattr(nbr_edge_x, 'eids') else tuple(nbr_edge_x))
        try:
            return tuple((t.data_ptr(), t._version, t.shape[0]) for t in (*nbr_nids, *nbr_edge_time, *feats) if t is not None)
        except RuntimeError:
            # inference tensors keep no version counter ("Inference tensors do not track version counter"): an in-place change could
            # not be seen, so such a call gets a stamp that matches nothing -- the row-per-slot computation, never the compact one
            return (object(),)

    def matches(self, nbr_nids, nbr_edge_time, nbr_edge_x=None) -> bool:
        try:
            return self._of(nbr_nids, nbr_edge_time, nbr_edge_x) == self.stamp
        except (AttributeError, IndexError, TypeError, RuntimeError):
            return False


class SampledHops(list):
    """A per-hop list (``seed_times``, ``nbr_nids``, ``nbr_edge_time``) as one sampler call published it: a plain ``list`` + the call's tag."""

    def __init__(self, items=(), tag: Optional[SamplerCallTag] = None) -> None:
        super().__init__(items)
        self.tag = tag

    def copy(self) -> 'SampledHops':
        return SampledHops(self, self.tag)


class EdgeFeaturesById(list):
    """``len(num hops)`` list of ``[S_h, k_h, D]`` tensors, backed by ``eids[h]`` (int32, -1 = pad) and ``table`` ([E, D])."""

    def __init__(self, eids: List[Tensor], table: Tensor, tag: Optional[SamplerCallTag] = None) -> None:
        super().__init__([None] * len(eids))
        self.eids, self.table = list(eids), table
        self.tag = tag

    def _dense(self, h: int) -> Tensor:
        got: Optional[Tensor] = super().__getitem__(h)
        if got is None:
            eid = self.eids[h]
            rows = self.table.index_select(0, eid.reshape(-1).clamp(min=0).long()).view(*eid.shape, self.table.shape[1])
            got = rows * (eid >= 0).unsqueeze(-1).to(rows.dtype)  # pad slots: zeros, like the dense copy
            super().__setitem__(h, got)
        return got

    def __getitem__(self, h):  # type: ignore[override]
        if isinstance(h, slice):
            return [self._dense(i) for i in range(*h.indices(len(self)))]
        return self._dense(h if h >= 0 else len(self) + h)

    def __iter__(self):
        return (self._dense(h) for h in range(len(self)))

    def dense(self) -> List[Tensor]:
        return [self._dense(h) for h in range(len(self))]

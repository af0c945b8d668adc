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


class EdgeFeaturesById(list):
    """``len(num hops)`` list of ``[S_h, k_h, D]`` tensors, backed by ``eids[h]`` (int32, -1 = pad) and ``table`` ([E, D])."""

    def __init__(self, eids: List[Tensor], table: Tensor) -> None:
        super().__init__([None] * len(eids))
        self.eids, self.table = list(eids), table

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

"""Unique node ids of a batch + global->local index map (tgm/hooks/dedup.py:17-67).

Used by the TGN loop: ``unique_nids`` = sorted unique of edge endpoints, extra
seed attributes and every valid (non-pad) sampled neighbor id;
``global_to_local(x)`` = position of ``x`` in ``unique_nids`` (int32).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from ..constants import PADDED_NODE_ID
from ..core import DGBatch, DGraph
from .base import SeedableHook, StatelessHook
from .registry import hook


@hook
class DeduplicationHook(StatelessHook, SeedableHook):
    """Deduplicate node IDs from batch fields and create index mappings to unique node embeddings.

    Key words: unique nodes, node ID mapper.
    """

    _cls_requires = {'edge_src', 'edge_dst'}
    _cls_produces = {'unique_nids', 'global_to_local'}

    def __init__(self, seed_nodes_keys: Optional[List[str]] = None, id: Optional[str] = None) -> None:
        super().__init__()
        self._id = id
        self.seed_keys = seed_nodes_keys
        self.__post_init__()

    def __call__(self, dg: DGraph, batch: DGBatch) -> DGBatch:
        device = batch.edge_src.device
        parts = [batch.edge_src, batch.edge_dst]
        for attr in self.requires:
            if not hasattr(batch, attr):
                raise ValueError(f'Missing seed node attribute {attr}')
            if 'nbr_nids' in attr:
                for hop_ids in getattr(batch, attr):
                    flat = hop_ids.reshape(-1)
                    parts.append(flat[flat != PADDED_NODE_ID].to(device))
            else:
                value = getattr(batch, attr)
                if value is not None:
                    parts.append(value)
        unique_nids = torch.unique(torch.cat(parts, dim=0), sorted=True)
        self.add_batch_attribute(batch, 'unique_nids', unique_nids)
        self.add_batch_attribute(batch, 'global_to_local', lambda x: torch.searchsorted(unique_nids, x).int())
        return batch

"""``SampledEdgeListHook`` (ours) -- the sampled neighbors of one hop as a compact edge list.

The reference's TGN training loop (examples/linkproppred/tgn.py:80-92) assembles, per batch and from a dozen torch ops with
two boolean-mask gathers, the edge list its ``GraphAttentionEmbedding`` consumes:

    mask = nbr != -1
    edge_index = stack([global_to_local(seeds.repeat_interleave(k)[mask]), global_to_local(nbr[mask])])
    edge_time  = nbr_edge_time[hop].flatten()[mask];   edge_x = nbr_edge_x[hop].flatten(0, -2)[mask]

This hook produces exactly those three tensors (``sampled_edge_index`` [2, E] int64, ``sampled_edge_time`` [E] int64,
``sampled_edge_x`` [E, D] float32; bit-identical, slot order) with ``tgmx_tgn_edge_list``: two launches, working from the
``DeduplicationHook``'s device-side result, so the only host wait of the whole chain is the read of the two sizes (unique
ids, E) -- which ``DGDataLoader(prefetch=1)`` moves behind the next batch's enqueue.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..core import DGBatch, DGraph
from .base import StatelessHook
from .registry import hook


@hook
class SampledEdgeListHook(StatelessHook):
    """Compact (edge_index, edge_time, edge_x) of the valid sampled neighbors of one hop, with deduplicated local node ids.

    Key words: edge index, compaction, TGN, graph attention embedding.
    """

    _cls_requires = {'seed_nids', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x', 'unique_nids'}
    _cls_produces = {'sampled_edge_index', 'sampled_edge_time', 'sampled_edge_x'}
    # consumed through the dedup hook's device-side handoff (batch._unique_dev): a pending finalizer for it need not run first
    _device_side_requires = frozenset({'unique_nids'})

    def __init__(self, hop: int = 0, id: Optional[str] = None) -> None:
        super().__init__()
        self.hop = int(hop)
        self._id = id
        self._ring = None
        self._turn = 0
        self.__post_init__()

    def __call__(self, dg: DGraph, batch: DGBatch) -> DGBatch:
        from ..nn.tgn import _edge_list_enqueue

        ei, et, ex, count = _edge_list_enqueue(batch, self.hop)
        dev = count.device
        if self._ring is None:
            self._ring = [(torch.zeros(1, dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(4)]  # one per batch in flight
        pin, ev = self._ring[self._turn % len(self._ring)]
        self._turn += 1
        with torch.cuda.device(dev):
            pin.copy_(count, non_blocking=True)
            ev.record()

        def finish() -> None:
            ev.synchronize()
            E = int(pin[0])
            self.add_batch_attribute(batch, 'sampled_edge_index', ei[:, :E])
            self.add_batch_attribute(batch, 'sampled_edge_time', et[:E])
            self.add_batch_attribute(batch, 'sampled_edge_x', ex[:E])

        batch._defer(finish, self.produces)
        return batch

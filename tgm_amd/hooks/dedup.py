"""Unique node ids of a batch + global->local index map (tgm/hooks/dedup.py:17-67).

Used by the TGN loop: ``unique_nids`` = sorted unique of edge endpoints, extra
seed attributes and every valid (non-pad) sampled neighbor id;
``global_to_local(x)`` = position of ``x`` in ``unique_nids`` (int32).

On a ROCm device the ids go through ``tgmx_unique_ids`` (bitmap over the node ids: no sort, pads skipped inside the
kernel instead of one boolean-mask sync per hop; ``csrc/dedup.hip``).  Host tensors keep the reference's own torch
formulation (``cat`` + ``torch.unique``) -- this hook is data plumbing with no device requirement in the reference.
"""
from __future__ import annotations

from typing import List, Optional

import ctypes

import torch

from .. import _native
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
        nbr_parts = []
        for attr in self.requires:
            if not hasattr(batch, attr):
                raise ValueError(f'Missing seed node attribute {attr}')
            if 'nbr_nids' in attr:
                nbr_parts += [hop_ids.reshape(-1) for hop_ids in getattr(batch, attr)]
            else:
                value = getattr(batch, attr)
                if value is not None:
                    parts.append(value)
        if device.type == 'cuda':
            self._unique_device(dg, batch, parts + [p.to(device) for p in nbr_parts], device)
            return batch
        flat = parts + [p[p != PADDED_NODE_ID] for p in nbr_parts]
        self._publish(batch, torch.unique(torch.cat(flat, dim=0), sorted=True))
        return batch

    def _publish(self, batch: DGBatch, unique_nids: torch.Tensor) -> None:
        self.add_batch_attribute(batch, 'unique_nids', unique_nids)
        self.add_batch_attribute(batch, 'global_to_local', lambda x: torch.searchsorted(unique_nids, x).int())

    def _unique_device(self, dg: DGraph, batch: DGBatch, parts: List[torch.Tensor], device: torch.device) -> None:
        """Enqueue ``tgmx_unique_ids`` and publish ``unique_nids`` / ``global_to_local`` once the count is on the host: right away
        (one device -> host wait, like ``torch.unique``), or -- for a batch the loader prefetches -- in the batch's finalizer."""
        lib = _native.load()
        parts = [p for p in parts if p.numel()]
        dtype = parts[0].dtype if parts else torch.int32
        parts = [p if (p.dtype == torch.int32 and p.is_contiguous()) else p.to(torch.int32).contiguous() for p in parts]
        if not parts:
            self._publish(batch, torch.empty(0, dtype=dtype, device=device))
            return
        if len(parts) > 16:
            parts = parts[:15] + [torch.cat(parts[15:])]
        N = int(dg._storage.num_nodes_global)
        total = sum(p.numel() for p in parts)
        ws = getattr(self, '_ws', None)
        if ws is None or ws[0] != (device, N):
            need = int(lib.tgmx_unique_ids_workspace_bytes(N))
            ring = [(torch.zeros(2, dtype=torch.int64, device=device), torch.zeros(2, dtype=torch.int64).pin_memory(), torch.cuda.Event())
                    for _ in range(4)]  # (count | status) on the device, its pinned host mirror, "mirror written" -- one set per batch in flight
            ws = self._ws = ((device, N), torch.zeros(need, dtype=torch.uint8, device=device), ring)  # zeros: the bitmap cleans itself
            self._turn = 0
        _, work, ring = ws
        cs, pin, ev = ring[self._turn % len(ring)]
        self._turn += 1
        count, status = cs[0:1], cs[1:2].view(torch.int32)[0:1]
        out = torch.empty(min(total, N), dtype=torch.int32, device=device)
        n = len(parts)
        ptrs = (ctypes.c_void_p * n)(*[p.data_ptr() for p in parts])
        sizes = (ctypes.c_int64 * n)(*[p.numel() for p in parts])
        with torch.cuda.device(device):
            _native.check(lib.tgmx_unique_ids(ptrs, sizes, n, N, work.data_ptr(), out.data_ptr(), count.data_ptr(), status.data_ptr(),
                                              _native.stream_ptr(device.index)), 'tgmx_unique_ids')  # fmt: skip
            pin.copy_(cs, non_blocking=True)
            ev.record()
        # device-side handoff for hooks further down the chain (they need neither the host count nor the finalized slice)
        batch.__dict__['_unique_dev'] = (out, count)

        def finish() -> None:
            ev.synchronize()  # the only wait: the result's size (torch.unique has the same one)
            cnt, st = pin.tolist()
            if st & 0xFFFFFFFF:
                cs[1].zero_()
                raise ValueError(f'node ids must satisfy 0 <= x < {N} (or -1 for a padded neighbor slot)')
            res = out[:cnt]
            self._publish(batch, res if dtype == torch.int32 else res.to(dtype))

        batch._defer(finish, self.produces)

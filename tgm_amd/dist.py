"""One process per GPU over RCCL/xGMI -- only what the hot path needs.

Sampling and TGAT aggregation shard with ZERO data-path communication: the
time-sorted stream is replicated in every GPU's HBM and each rank takes a
contiguous slice of every global batch's edges as its seeds
(:class:`EdgeShardHook`).  In streaming (ring) mode every rank still applies the
whole batch's update to its replica of the rings (2 * batch_size records:
microseconds), so replicas never diverge; in csr mode there is no state at all.
Concatenating the ranks' outputs in rank order reproduces the single-GPU result
up to the documented row permutation.  The only collective on the path is the
TGN memory all-gather (``tgm_amd.nn.tgn``).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from .core import DGBatch, DGraph
from .hooks.base import StatelessHook
from .hooks.registry import hook


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1 process if unset)."""
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Join the job (``nccl`` == RCCL on ROCm when a GPU is visible, else ``gloo``)."""
    import torch.distributed as dist

    rank, world, local = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # TGMX_DIST_BACKEND=gloo: functional check of the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get('TGMX_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of ``n`` items for ``rank``."""
    return (n * rank) // world, (n * (rank + 1)) // world


@hook
class EdgeShardHook(StatelessHook):
    """Publish this rank's contiguous slice of the batch's edges as ``shard_src``,
    ``shard_dst``, ``shard_time`` (views) and ``shard_lo`` (its offset in the batch).

    Key words: data parallel, shard, rank.
    """

    _cls_requires = {'edge_src', 'edge_dst', 'edge_time'}
    _cls_produces = {'shard_src', 'shard_dst', 'shard_time', 'shard_lo'}

    def __init__(self, rank: int, world_size: int, id: Optional[str] = None) -> None:
        super().__init__()
        if not 0 <= rank < world_size:
            raise ValueError(f'rank {rank} outside [0, {world_size})')
        self.rank, self.world_size = rank, world_size
        self._id = id
        self.__post_init__()

    def __call__(self, dg: DGraph, batch: DGBatch) -> DGBatch:
        src = batch.edge_src
        lo, hi = shard_bounds(src.shape[0], self.rank, self.world_size)
        n = hi - lo
        self.add_batch_attribute(batch, 'shard_src', src.narrow(0, lo, n))  # narrow: half the cost of [lo:hi]
        self.add_batch_attribute(batch, 'shard_dst', batch.edge_dst.narrow(0, lo, n))
        self.add_batch_attribute(batch, 'shard_time', batch.edge_time.narrow(0, lo, n))
        self.add_batch_attribute(batch, 'shard_lo', lo)
        return batch

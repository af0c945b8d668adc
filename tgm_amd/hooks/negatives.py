"""Random negative destinations for link prediction.

Same contract as tgm/hooks/negatives/sampler.py:15-65: ``neg`` = uniform int32
ids in ``[low, high)`` (one per positive edge, times ``neg_ratio``), ``neg_time``
= a copy of the batch's edge times; it produces the third seed group the neighbor sampler consumes.
On a ROCm device both outputs come from one ``tgmx_random_negatives`` launch (counter-based generator seeded from
``seed`` or ``torch.initial_seed()``: reproducible under ``torch.manual_seed``, but not torch's Philox stream); host
tensors use ``torch.randint`` like the reference.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _native
from ..core import DGBatch, DGraph
from .base import StatelessHook
from .registry import hook


@hook
class RandomNegativeEdgeSamplerHook(StatelessHook):
    """Random sampling of negative edges for dynamic link prediction.

    Key words: negative sampler, random, uniform, training, link prediction.
    """

    _cls_requires = {'edge_src', 'edge_dst', 'edge_time'}
    _cls_produces = {'neg', 'neg_time'}

    def __init__(
        self,
        low: int,
        high: int,
        neg_ratio: float = 1.0,
        id: Optional[str] = None,
        seed: Optional[int] = None,
        like: str = 'edge_dst',
        time_key: str = 'edge_time',
    ) -> None:
        super().__init__()
        if not 0 < neg_ratio <= 1:
            raise ValueError(f'neg_ratio must be in (0, 1], got: {neg_ratio}')
        if not low < high:
            raise ValueError(f'low ({low}) must be strictly less than high ({high})')
        self.low, self.high, self.neg_ratio = low, high, neg_ratio
        self._seed = seed
        self._gen: Optional[torch.Generator] = None
        self._rng_seed: Optional[int] = None  # device generator: (seed, call counter, element index)
        self._calls = 0
        # extensions (defaults = the reference's behaviour): draw one negative per element of
        # batch.<like> and copy batch.<time_key>; used to sample for a rank's shard of the batch
        self._like, self._time_key = like, time_key
        self._id = id
        self.__post_init__()
        self._requires |= {like, time_key}

    def __call__(self, dg: DGraph, batch: DGBatch) -> DGBatch:
        n = round(self.neg_ratio * getattr(batch, self._like).size(0))
        device = dg.device
        if n == 0:
            neg = torch.empty((0,), dtype=torch.int32, device=device)
            neg_time = torch.empty((0,), dtype=torch.int64, device=device)
        elif device.type == 'cuda':
            t_in = getattr(batch, self._time_key)
            if t_in.dtype != torch.int64 or not t_in.is_contiguous():
                t_in = t_in.to(torch.int64).contiguous()
            ne = getattr(self, '_new_empty', None)
            if ne is None or ne[0] != device:  # bound new_empty of per-dtype prototypes: cheaper than torch.empty(dtype=, device=)
                ne = self._new_empty = (device, torch.empty(0, dtype=torch.int32, device=device).new_empty,
                                        torch.empty(0, dtype=torch.int64, device=device).new_empty)
            neg = ne[1](n)
            neg_time = ne[2](t_in.shape[0])
            if self._rng_seed is None:
                self._rng_seed = (self._seed if self._seed is not None else torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
            self._calls += 1
            # a rank's share of a sharded batch (like='shard_dst', tgm_amd.dist.EdgeShardHook): its draws are the slice
            # [shard_lo, shard_lo + n) of the ids one rank would draw for the whole batch
            index0 = int(getattr(batch, 'shard_lo', 0) or 0) if self._like != 'edge_dst' else 0
            rc = _native.load().tgmx_random_negatives_at(self.low, self.high, n, self._rng_seed, self._calls, index0, neg.data_ptr(), t_in.data_ptr(),
                                                         t_in.shape[0], neg_time.data_ptr(), _native.stream_ptr(device.index))  # fmt: skip
            if rc:
                _native.check(rc, 'tgmx_random_negatives_at')
        else:
            gen = None
            if self._seed is not None:
                if self._gen is None:
                    self._gen = torch.Generator(device=device)
                    self._gen.manual_seed(self._seed)
                gen = self._gen
            neg = torch.randint(self.low, self.high, (n,), dtype=torch.int32, device=device, generator=gen)
            neg_time = getattr(batch, self._time_key).clone()
        self.add_batch_attribute(batch, 'neg', neg)
        self.add_batch_attribute(batch, 'neg_time', neg_time)
        return batch

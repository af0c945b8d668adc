"""CPU restatement of the reference's uniform neighbor sampler -- TEST INFRASTRUCTURE ONLY.

Follows tgm/hooks/neighbors/uniform.py:87-142 (hop loop: hop h > 0 seeds = hop h-1 outputs flattened, pads included;
candidates = every edge strictly before the batch's first timestamp, `DGSliceTracker(end_time=min(edge_time) - 1)`)
and tgm/core/_storage/backends/array_backend.py:108-171 (`get_nbrs`: per unique seed node the list of (event, nbr) in
event order -- source role before destination role for a self loop; `random.sample` when there are more than k;
left-aligned rows padded with (-1, 0, 0.0); every occurrence of a node in the hop's seed list gets the same row).

Pinned to the reference: tests/golden/g11_uniform_*.npz were produced by the reference hook with Python's `random`
seeded (4242); this restatement consumes `random` in the same order, so it reproduces the sampled rows exactly
(tests/test_uniform_oracle_cpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.
"""
from __future__ import annotations

import random
from typing import List, Optional, Sequence, Tuple

import numpy as np


def get_nbrs(src: np.ndarray, dst: np.ndarray, ts: np.ndarray, edge_x: Optional[np.ndarray], seeds: np.ndarray, k: int, ev_hi: int,
             directed: bool, rng=random) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:  # fmt: skip
    """array_backend.py:108-171 on the first ev_hi edges of the (time-sorted) stream."""
    D = 0 if edge_x is None else edge_x.shape[1]
    uniq, inverse = np.unique(seeds, return_inverse=True)  # torch.unique sorts as well
    nbrs = {int(n): [] for n in uniq}
    for i in range(ev_hi):
        s, d = int(src[i]), int(dst[i])
        if s in nbrs:
            nbrs[s].append((i, d))
        if not directed and d in nbrs:
            nbrs[d].append((i, s))
    S = len(seeds)
    out_n = np.full((S, k), -1, np.int32)
    out_t = np.zeros((S, k), np.int64)
    out_x = np.zeros((S, k, D), np.float32)
    for i, node in enumerate(uniq.tolist()):
        cand = nbrs[node]
        if not cand:
            continue
        if len(cand) > k:
            cand = rng.sample(cand, k=k)
        m = len(cand)
        rows = inverse == i
        out_n[rows, :m] = np.asarray([c[1] for c in cand], np.int32)
        out_t[rows, :m] = np.asarray([ts[c[0]] for c in cand], np.int64)
        if D:
            out_x[rows, :m] = np.stack([edge_x[c[0]] for c in cand])
    return out_n, out_t, out_x


def step(src, dst, ts, edge_x, seeds: np.ndarray, seed_times: np.ndarray, num_nbrs: Sequence[int], batch_tmin: int, directed: bool,
         rng=random) -> List[Tuple[np.ndarray, ...]]:  # fmt: skip
    """uniform.py:87-142 for one batch: [(seed_nids, seed_times, nbr_nids, nbr_edge_time, nbr_edge_x)] per hop."""
    ev_hi = int(np.searchsorted(ts, batch_tmin, side='left'))  # events with time <= tmin - 1
    hops = []
    cur_n, cur_t = seeds, seed_times
    for k in num_nbrs:
        n, t, x = get_nbrs(src, dst, ts, edge_x, cur_n, k, ev_hi, directed, rng)
        hops.append((cur_n, cur_t, n, t, x))
        cur_n, cur_t = n.reshape(-1), t.reshape(-1)
    return hops

"""ORACLE (test infrastructure only -- never imported by the product path).

``RingSamplerCPU``: a torch-CPU restatement of the reference's ring-buffer
sampler as the *tensor program* it is, including its arithmetic quirks, so it
(a) reproduces the imported reference bit-for-bit at every scale and (b) can be
timed on the GPU box's host cores as the "reference algorithm on CPU" baseline
(``bench.py`` ``cpu_baseline``, kind "port").

Follows /root/reference/tgm/hooks/neighbors/recency.py:
  :93-102   state: ids[N,B] int32 = -1, times[N,B] int64 = 0, feats[N,B,D] f32 = 0,
            write_pos[N]
  :239-321  lookup: unroll ring oldest->newest, mask ``time < q`` & non-pad,
            rightmost valid position, window of k ending there, pads left
  :323-399  update: cat[src-role, dst-role] entries; stable argsort of the
            composite key; per-run "keep last B"; scatter at
            (write_pos + rank) % B; write_pos += kept

Quirk preserved on purpose (``key_arith='int32'``, the reference's behaviour):
recency.py:347 computes ``node_ids * max_time`` with an int32 tensor and a 0-dim
int64 tensor; type promotion keeps int32, so the product WRAPS whenever
node * (max_time+1) >= 2**31 (always, at tgbl-wiki scale).  Entries of one node
are then no longer contiguous after the sort, ``unique_consecutive`` splits them
into several runs, runs of the same node collide on ring slots (last write
wins) while write_pos advances by the total -- leaving stale/empty slots inside
the ring.  ``key_arith='int64'`` gives the intended (node, time) order.

Parity status: pinned against tests/golden (all families incl. g3, which is in
the wrapping regime) by tests/test_oracle_golden.py.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

PAD_ID = -1


class RingSamplerCPU:
    def __init__(self, num_nodes: int, num_nbrs: Sequence[int], edge_dim: int, directed: bool = False, key_arith: str = 'int32'):
        assert key_arith in ('int32', 'int64')
        self.N, self.D = int(num_nodes), int(edge_dim)
        self.num_nbrs = list(num_nbrs)
        self.B = max(self.num_nbrs)
        self.directed = bool(directed)
        self.key_arith = key_arith
        self.ids = torch.empty((self.N, self.B), dtype=torch.int32)
        self.times = torch.empty((self.N, self.B), dtype=torch.int64)
        self.feats = torch.empty((self.N, self.B, self.D), dtype=torch.float32)
        self.wpos = torch.empty(self.N, dtype=torch.int64)
        self.reset()

    def reset(self) -> None:
        self.ids.fill_(PAD_ID)
        self.times.zero_()
        self.feats.zero_()
        self.wpos.zero_()

    # ------------------------------------------------------------------
    def lookup(self, nodes: Tensor, q: Tensor, k: int) -> Tuple[Tensor, Tensor, Tensor]:
        B = self.B
        rows = nodes.long()  # -1 (pad seed) indexes the last row, as in the reference
        S = rows.numel()
        ar = torch.arange(B)
        slots = (self.wpos[rows][:, None] + ar[None, :]) % B  # unrolled position -> ring slot
        row_ids = self.ids[rows]
        row_t = self.times[rows]
        t_un = row_t.gather(1, slots)
        id_un = row_ids.gather(1, slots)
        ok = (t_un < q[:, None]) & (id_un != PAD_ID)
        # rightmost True per row, -1 when none
        last = torch.where(ok.any(1), (ok * ar[None, :]).amax(1), torch.full((S,), -1, dtype=torch.int64))
        pos = last[:, None] - torch.arange(k - 1, -1, -1)[None, :]  # [S, k] unrolled positions, <0 = none
        have = pos >= 0
        slot_k = slots.gather(1, pos.clamp(min=0))
        out_i = torch.where(have, row_ids.gather(1, slot_k), torch.full((), PAD_ID, dtype=torch.int32))
        out_t = torch.where(have, row_t.gather(1, slot_k), torch.zeros((), dtype=torch.int64))
        out_x = self.feats[rows].gather(1, slot_k[:, :, None].expand(-1, -1, self.D))
        out_x = out_x.masked_fill(~have[:, :, None], 0.0)
        return out_i, out_t, out_x

    # ------------------------------------------------------------------
    def update(self, src: Tensor, dst: Tensor, ts: Tensor, edge_x: Optional[Tensor]) -> None:
        B = self.B
        n = src.numel()
        if edge_x is None:
            edge_x = torch.zeros((n, self.D), dtype=torch.float32)
        if self.directed:
            node, nbr, t, x = src, dst, ts, edge_x
        else:
            node, nbr = torch.cat([src, dst]), torch.cat([dst, src])
            t, x = torch.cat([ts, ts]), torch.cat([edge_x, edge_x])
        span = t.max() + 1
        if self.key_arith == 'int32':
            # int32 tensor * 0-dim int64 tensor -> int32 (wraps), then + int64 times
            key = (node.to(torch.int32) * span.to(torch.int32)).to(torch.int64) + t
        else:
            key = node.to(torch.int64) * span + t
        order = torch.argsort(key, stable=True)
        node_s, nbr_s, t_s, x_s = node[order].long(), nbr[order], t[order], x[order]

        # runs of consecutive equal node ids in sorted order
        m = node_s.numel()
        new_run = torch.ones(m, dtype=torch.bool)
        new_run[1:] = node_s[1:] != node_s[:-1]
        run_id = torch.cumsum(new_run, 0) - 1
        run_len = torch.bincount(run_id)
        run_first = torch.cumsum(run_len, 0) - run_len
        rank = torch.arange(m) - run_first[run_id]
        drop = (run_len[run_id] - B).clamp(min=0)
        keep = rank >= drop
        node_k, nbr_k, t_k, x_k = node_s[keep], nbr_s[keep], t_s[keep], x_s[keep]
        off_k = (rank - drop)[keep]
        slot = (self.wpos[node_k] + off_k) % B
        # Duplicate (node, slot) targets (runs of one node colliding): the reference's index_put_ (recency.py:381-395)
        # runs serially -- last assignment wins -- up to 3000 scattered elements; above that torch parallelises the
        # scatter and the winner is a thread race, i.e. unspecified.  The restatement defines last-wins at every size
        # (numpy fancy assignment is sequential), which is what the kernels implement.
        nk, sl = node_k.numpy(), slot.numpy()
        self.ids.numpy()[nk, sl] = nbr_k.to(torch.int32).numpy()
        self.times.numpy()[nk, sl] = t_k.numpy()
        self.feats.numpy()[nk, sl] = x_k.numpy()
        self.wpos += torch.bincount(node_k, minlength=self.N)

    # ------------------------------------------------------------------
    def step(self, seeds: Tensor, seed_times: Tensor, src: Tensor, dst: Tensor, ts: Tensor, edge_x: Optional[Tensor]):
        """One hook call: all hops, then the update (skipped when there are no seeds)."""
        if seeds.numel() == 0:
            return None
        hops: List[Tuple[Tensor, ...]] = []
        cur_n, cur_t = seeds, seed_times
        for k in self.num_nbrs:
            o_i, o_t, o_x = self.lookup(cur_n, cur_t, k)
            hops.append((cur_n, cur_t, o_i, o_t, o_x))
            cur_n, cur_t = o_i.reshape(-1), o_t.reshape(-1)
        if src.numel():
            self.update(src, dst, ts, edge_x)
        return hops

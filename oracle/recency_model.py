"""ORACLE (test infrastructure only -- never imported by the product path).

Plain-Python / numpy restatement of the reference's k-most-recent temporal
neighbor sampler, written from the behavioural model in SURVEY.md Appendix A.

Follows (reference, paths relative to /root/reference):
  * tgm/hooks/neighbors/recency.py:93-102   state (ring of B = max(num_nbrs))
  * tgm/hooks/neighbors/recency.py:119-171  hop loop, update-after-lookup,
                                            no-seed batches skip the update
  * tgm/hooks/neighbors/recency.py:239-321  lookup: strict ``t < q``, rightmost
                                            valid entry, k-window, left padding
  * tgm/hooks/neighbors/recency.py:323-399  update: stable (node, time) order
                                            over cat[src-role, dst-role]

Two independent formulations are provided so they can be checked against
each other and against the golden vectors captured from the reference:

  ``HistoryModel``  -- per-node append-only python lists (the abstract model,
                       obviously-correct, slow: small cases only)
  ``CsrModel``      -- a static per-node index sorted by
                       (batch_idx, time, role, eid) queried with
                       (epoch_start, batch_start) event bounds (Appendix A.3);
                       this is the formulation the HIP ``csr`` path implements.

Parity status: pinned -- see tests/test_oracle_golden.py, which checks both
models against tests/golden/*.npz (outputs of the imported reference).
"""
from __future__ import annotations

import bisect
from typing import List, Sequence, Tuple

import numpy as np

PAD_ID = -1  # tgm/constants.py:3


def _empty_out(S: int, k: int, D: int):
    return (
        np.full((S, k), PAD_ID, dtype=np.int32),
        np.zeros((S, k), dtype=np.int64),
        np.zeros((S, k, D), dtype=np.float32),
    )


class HistoryModel:
    """Streaming model: H[n] = chronological list of (nbr, time, eid)."""

    def __init__(self, num_nodes: int, num_nbrs: Sequence[int], directed: bool = False):
        self.N = int(num_nodes)
        self.num_nbrs = list(num_nbrs)
        self.B = max(self.num_nbrs)
        self.directed = bool(directed)
        self.reset()

    def reset(self) -> None:
        self.H: List[List[Tuple[int, int, int]]] = [[] for _ in range(self.N)]

    # -- lookup -----------------------------------------------------------
    def lookup(self, nodes, qtimes, k: int, edge_x: np.ndarray | None):
        nodes = np.asarray(nodes)
        qtimes = np.asarray(qtimes)
        S = len(nodes)
        D = 0 if edge_x is None else edge_x.shape[1]
        out_n, out_t, out_x = _empty_out(S, k, D)
        for s in range(S):
            n, q = int(nodes[s]), int(qtimes[s])
            if n < 0:
                # pad seeds (hop >= 1) carry q == 0: nothing is < 0, all-pad row
                continue
            win = self.H[n][-self.B:]
            j = -1
            for p in range(len(win) - 1, -1, -1):
                if win[p][1] < q:
                    j = p
                    break
            if j < 0:
                continue
            take = win[max(0, j - k + 1): j + 1]
            off = k - len(take)
            for c, (nbr, t, eid) in enumerate(take):
                out_n[s, off + c] = nbr
                out_t[s, off + c] = t
                if D:
                    out_x[s, off + c] = edge_x[eid]
        return out_n, out_t, out_x

    # -- update -----------------------------------------------------------
    def update(self, src, dst, ts, eid0: int) -> None:
        src = np.asarray(src)
        dst = np.asarray(dst)
        ts = np.asarray(ts)
        n = len(src)
        ent = [(int(src[i]), int(ts[i]), i, int(dst[i]), eid0 + i) for i in range(n)]
        if not self.directed:
            ent += [(int(dst[i]), int(ts[i]), n + i, int(src[i]), eid0 + i) for i in range(n)]
        # stable (node, time) order == sort by (node, time, position in cat)
        ent.sort(key=lambda e: (e[0], e[1], e[2]))
        for node, t, _, nbr, eid in ent:
            self.H[node].append((nbr, t, eid))
        for node in set(e[0] for e in ent):
            if len(self.H[node]) > self.B:
                del self.H[node][: len(self.H[node]) - self.B]

    # -- one hook call ----------------------------------------------------
    def step(self, seeds, seed_times, src, dst, ts, eid0: int, edge_x):
        """Returns per-hop lists (seed_nids, seed_times, nbr_nids, nbr_times, nbr_x)."""
        hops = []
        seeds = np.asarray(seeds, dtype=np.int32)
        seed_times = np.asarray(seed_times, dtype=np.int64)
        if len(seeds) == 0:
            return None  # reference emits empties and skips the update
        cur_n, cur_t = seeds, seed_times
        for k in self.num_nbrs:
            o_n, o_t, o_x = self.lookup(cur_n, cur_t, k, edge_x)
            hops.append((cur_n, cur_t, o_n, o_t, o_x))
            cur_n, cur_t = o_n.reshape(-1), o_t.reshape(-1)
        if len(src):
            self.update(src, dst, ts, eid0)
        return hops


class CsrModel:
    """Static per-node index, order key (batch_idx, time, role, eid).

    ``batch_starts`` are the event indices at which the loader starts a batch
    (strictly increasing, first == first event, implicit end == E).
    Lookup for a batch starting at event ``ev_hi`` in an epoch that started at
    event ``ev_lo`` sees node n's entries with ``ev_lo <= eid < ev_hi``; only
    the last B of them are observable.
    """

    def __init__(self, src, dst, ts, num_nodes: int, batch_starts, directed: bool = False):
        src = np.asarray(src, dtype=np.int64)
        dst = np.asarray(dst, dtype=np.int64)
        ts = np.asarray(ts, dtype=np.int64)
        E = len(src)
        bstarts = np.asarray(batch_starts, dtype=np.int64)
        bidx = np.searchsorted(bstarts, np.arange(E), side='right') - 1
        node = [src]
        nbr = [dst]
        role = [np.zeros(E, dtype=np.int64)]
        if not directed:
            node.append(dst)
            nbr.append(src)
            role.append(np.ones(E, dtype=np.int64))
        reps = len(node)
        node = np.concatenate(node)
        nbr = np.concatenate(nbr)
        role = np.concatenate(role)
        eid = np.tile(np.arange(E, dtype=np.int64), reps)
        t = np.tile(ts, reps)
        b = np.tile(bidx, reps)
        order = np.lexsort((eid, role, t, b, node))
        self.adj_nbr = nbr[order].astype(np.int32)
        self.adj_ts = t[order]
        self.adj_eid = eid[order].astype(np.int32)
        cnt = np.bincount(node, minlength=num_nodes)
        self.indptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        self.N = num_nodes

    def lookup(self, nodes, qtimes, k: int, B: int, ev_lo: int, ev_hi: int, edge_x):
        nodes = np.asarray(nodes)
        qtimes = np.asarray(qtimes)
        S = len(nodes)
        D = 0 if edge_x is None else edge_x.shape[1]
        out_n, out_t, out_x = _empty_out(S, k, D)
        for s in range(S):
            n, q = int(nodes[s]), int(qtimes[s])
            if n < 0:
                continue
            a, z = int(self.indptr[n]), int(self.indptr[n + 1])
            seg_eid = self.adj_eid[a:z]
            # entries of earlier batches form a prefix (ev_hi is a batch boundary)
            p_hi = a + int(np.count_nonzero(seg_eid < ev_hi))
            p_lo = a + int(np.count_nonzero(seg_eid < ev_lo))
            w_lo = max(p_lo, p_hi - B)
            # inside the window times are non-decreasing -> valid set is a prefix
            tw = self.adj_ts[w_lo:p_hi]
            cnt = bisect.bisect_left(tw.tolist(), q)
            j = w_lo + cnt  # one past the rightmost valid entry
            lo = max(w_lo, j - k)
            m = j - lo
            if m <= 0:
                continue
            out_n[s, k - m:] = self.adj_nbr[lo:j]
            out_t[s, k - m:] = self.adj_ts[lo:j]
            if D:
                out_x[s, k - m:] = edge_x[self.adj_eid[lo:j]]
        return out_n, out_t, out_x

    def step(self, seeds, seed_times, num_nbrs, ev_lo: int, ev_hi: int, edge_x):
        B = max(num_nbrs)
        cur_n = np.asarray(seeds, dtype=np.int32)
        cur_t = np.asarray(seed_times, dtype=np.int64)
        hops = []
        for k in num_nbrs:
            o_n, o_t, o_x = self.lookup(cur_n, cur_t, k, B, ev_lo, ev_hi, edge_x)
            hops.append((cur_n, cur_t, o_n, o_t, o_x))
            cur_n, cur_t = o_n.reshape(-1), o_t.reshape(-1)
        return hops

"""Synthetic temporal edge streams shaped like the TGB datasets named in BASELINE.json.

No dataset can be downloaded here, so the benchmark, the golden-vector
generator and the parity tests all draw their inputs from this one seeded
generator (SURVEY.md section 8(d) fixes the shapes).  Streams are produced
with a ``torch.Generator`` on the requested device: the CPU stream for a
given (shape, seed) is bit-stable and is what the golden fixtures use; the
on-device stream is used for the large shapes, where every rank regenerates
the same replicated stream from the same seed.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

# name -> (n_src, n_dst, bipartite, num_edges, t_lo, t_hi, edge_dim, zipf_a)
_SHAPES: Dict[str, dict] = {
    # tgbl-wiki: users -> pages, ~31 days of seconds, 172-d edge features
    'wiki': dict(n_src=8227, n_dst=1000, bipartite=True, E=157_474, t_lo=0, t_hi=2_678_373, D=172),
    # tgbl-review: users -> items, unix-scale seconds
    'review': dict(n_src=300_000, n_dst=50_000, bipartite=True, E=4_800_000, t_lo=900_000_000, t_hi=1_500_000_000, D=16),
    # tgbl-comment: user <-> user (non-bipartite)
    'comment': dict(n_src=1_000_000, n_dst=1_000_000, bipartite=False, E=44_000_000, t_lo=1_100_000_000, t_hi=1_300_000_000, D=16),
}


@dataclass
class EdgeStream:
    """A time-sorted COO edge stream (all tensors on one device)."""

    src: torch.Tensor  # [E] int32
    dst: torch.Tensor  # [E] int32
    ts: torch.Tensor  # [E] int64, non-decreasing
    edge_x: Optional[torch.Tensor]  # [E, D] float32 or None
    node_x: torch.Tensor  # [N, d0] float32 (static node features)
    num_nodes: int
    name: str

    @property
    def num_edges(self) -> int:
        return int(self.src.numel())

    @property
    def edge_dim(self) -> int:
        return 0 if self.edge_x is None else int(self.edge_x.shape[1])

    def to(self, device) -> 'EdgeStream':
        mv = lambda t: None if t is None else t.to(device)
        return EdgeStream(mv(self.src), mv(self.dst), mv(self.ts), mv(self.edge_x), mv(self.node_x), self.num_nodes, self.name)


def _zipf_draw(n_items: int, size: int, a: float, gen: torch.Generator, device) -> torch.Tensor:
    """Inverse-CDF draw from p(r) ~ 1 / (r+1)^a over ``n_items`` ranks, ranks shuffled."""
    w = 1.0 / torch.arange(1, n_items + 1, dtype=torch.float64, device=device).pow(a)
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    u = torch.rand(size, generator=gen, device=device, dtype=torch.float64)
    rank = torch.searchsorted(cdf, u).clamp_(max=n_items - 1)
    perm = torch.randperm(n_items, generator=gen, device=device)
    return perm[rank]


def make_stream(
    shape: str = 'wiki',
    seed: int = 1337,
    num_edges: Optional[int] = None,
    edge_dim: Optional[int] = None,
    node_dim: int = 1,
    device: str | torch.device = 'cpu',
    n_src: Optional[int] = None,
    n_dst: Optional[int] = None,
    t_hi: Optional[int] = None,
) -> EdgeStream:
    """Build a ``shape``-like stream.  Any of the size knobs may be overridden
    (the parity tests shrink E / D / node counts; the benchmark does not)."""
    cfg = dict(_SHAPES[shape])
    if num_edges is not None:
        cfg['E'] = int(num_edges)
    if edge_dim is not None:
        cfg['D'] = int(edge_dim)
    if n_src is not None:
        cfg['n_src'] = int(n_src)
    if n_dst is not None:
        cfg['n_dst'] = int(n_dst)
    if t_hi is not None:
        cfg['t_hi'] = int(t_hi)
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    E = cfg['E']

    ts = torch.randint(cfg['t_lo'], cfg['t_hi'] + 1, (E,), generator=gen, device=device, dtype=torch.int64)
    ts, _ = torch.sort(ts)
    if cfg['bipartite']:
        num_nodes = cfg['n_src'] + cfg['n_dst']
        src = _zipf_draw(cfg['n_src'], E, 0.6, gen, device)
        dst = _zipf_draw(cfg['n_dst'], E, 1.0, gen, device) + cfg['n_src']
    else:
        num_nodes = cfg['n_src']
        src = _zipf_draw(num_nodes, E, 0.8, gen, device)
        dst = _zipf_draw(num_nodes, E, 0.8, gen, device)
    D = cfg['D']
    edge_x = torch.rand((E, D), generator=gen, device=device, dtype=torch.float32) if D > 0 else None
    node_x = torch.randn((num_nodes, node_dim), generator=gen, device=device, dtype=torch.float32)
    return EdgeStream(src.to(torch.int32), dst.to(torch.int32), ts, edge_x, node_x, num_nodes, shape)

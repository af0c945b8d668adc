#!/usr/bin/env python
"""BASELINE config 5: tgbn-trade-like stream (255 nodes, yearly snapshots), DGData.discretize + one TGCN cell step per
snapshot; prints one JSON line.   python tools/bench_tgcn.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import DGData, DGDataLoader, DGraph  # noqa: E402
from tgm_amd.nn import TGCN  # noqa: E402

dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
N, E, year, Y = 255, 468_000, 365 * 24 * 3600, 30  # tgbn-trade: ~255 nodes, ~468k edges, 30+ yearly snapshots
ts = torch.from_numpy(np.sort(rng.integers(0, Y * year, E)))
ei = torch.from_numpy(rng.integers(0, N, (E, 2)).astype(np.int32))
raw = DGData.from_raw(ts, ei, torch.rand(E, 1), static_node_x=torch.randn(N, 16), time_delta='s')
torch.cuda.synchronize()
t0 = time.perf_counter()
data = raw.clone().to(dev) if hasattr(raw, 'to') else raw
data = raw.discretize('Y', device=dev)
torch.cuda.synchronize()
t_disc = time.perf_counter() - t0
dg = DGraph(data, device=dev)
cell = TGCN(16, 32).to(dev).eval()


def epoch():
    H = None
    n = 0
    with torch.no_grad():
        for batch in DGDataLoader(dg, batch_unit='Y'):
            H = cell(dg.static_node_x, torch.stack([batch.edge_src, batch.edge_dst]), None, H)
            n += 1
    return n


epoch()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    n = epoch()
torch.cuda.synchronize()
t1 = time.perf_counter()
# training step (round 4: the cell has a backward): forward with gradients + a linear head's loss + backward + Adam, per snapshot, the
# recurrent state detached between snapshots like examples/nodeproppred/tgcn.py:60-101
cell.train()
head = torch.nn.Linear(32, 8).to(dev)
opt = torch.optim.Adam(list(cell.parameters()) + list(head.parameters()), lr=1e-3)
target = torch.randn(N, 8, device=dev)


def train_epoch():
    H = None
    for batch in DGDataLoader(dg, batch_unit='Y'):
        opt.zero_grad(set_to_none=True)
        H = cell(dg.static_node_x, torch.stack([batch.edge_src, batch.edge_dst]), None, H)
        loss = ((head(torch.relu(H)) - target) ** 2).mean()
        loss.backward()
        opt.step()
        H = H.detach()
    return float(loss)


train_epoch()
torch.cuda.synchronize()
t2 = time.perf_counter()
for _ in range(reps):
    last_loss = train_epoch()
torch.cuda.synchronize()
t3 = time.perf_counter()
print(json.dumps({
    'tgcn_train_step_us_per_snapshot': 1e6 * (t3 - t2) / (reps * n), 'train_loss_last': last_loss,
    'what': 'BASELINE cfg5: tgbn-trade-like (255 nodes, 468k edges, 30 yearly snapshots): discretize + TGCN(16 -> 32) step per snapshot',
    'discretize_ms': 1e3 * t_disc, 'discretized_edges': int(data.edge_index.shape[0]), 'snapshots': n,
    'tgcn_step_us_per_snapshot': 1e6 * (t1 - t0) / (reps * n),
}))

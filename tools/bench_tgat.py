#!/usr/bin/env python
"""Time the TGAT eval forward at the headline batch shape (run on the GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tgm_amd.nn import TGAT  # noqa: E402
from tgm_amd.synth import make_stream  # noqa: E402

stream = make_stream('wiki', seed=1337)
dev = torch.device('cuda', 0)
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, 200, [20, 20], 'ring', dev)
enc = TGAT(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2).to(dev).eval()
starts = loader._starts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
node_x = dg.static_node_x
with hm.activate('bench'), torch.no_grad():
    for i in range(300):
        b = loader(starts[i])
    for _ in range(5):
        z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300, 300 + n):
        b = loader(starts[i])
        z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f'sampler + TGAT forward: host {1e6 * (t1 - t0) / n:.1f} us/step, total {1e6 * (t2 - t0) / n:.1f} us/step, z {tuple(z.shape)}')

#!/usr/bin/env python
"""Time the TGAT eval forward (and sampler + forward) at the headline batch shape on the GPU box;
prints one JSON line.  `python tools/bench_tgat.py [n_batches] [dense|by_id]` (by_id: the sampler publishes edge ids, the
attention reads the rows of the resident store -- RecencyNeighborHook(edge_features='by_id'))"""
import json
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
features = sys.argv[2] if len(sys.argv) > 2 else 'dense'
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, 200, [20, 20], 'ring', dev, pool=1, edge_features=features)  # the lowered chain into one persistent output set
enc = TGAT(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2).to(dev).eval()
starts = loader._starts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
node_x = dg.static_node_x
with hm.activate('bench'), torch.no_grad():
    for i in range(300):
        b = loader(starts[i])
    for _ in range(10):  # the first two forwards pay one-off costs (weight fold, code-object load: tens of ms)
        z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
        torch.cuda.synchronize()
    reps = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
        e1.record()
        torch.cuda.synchronize()
        reps.append(e0.elapsed_time(e1) / 20 * 1000)
    fwd_us = sorted(reps)[len(reps) // 2]  # median of 5 x 20 back-to-back forwards on one batch
    t0 = time.perf_counter()
    for i in range(300, 300 + n):
        b = loader(starts[i])
        z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # the same batches with the sampler on the loader's own stream, two batches ahead of the forward (DGDataLoader(side_stream=True): the
    # sampler's launches issued by the library's launch worker; three output sets) -- an HBM-bound gather beside an MFMA- / latency-bound forward
    from tgm_amd import DGDataLoader  # noqa: E402

    bs = 200
    def ahead(lo, hi):
        return DGDataLoader(dg.slice_events(lo * bs, hi * bs), batch_size=bs, hook_manager=hm, output_pool=3, prefetch=2, side_stream=True)

    first = 300 + n
    for b in ahead(first, first + 40):
        z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
    torch.cuda.synchronize()
    n2 = min(n, len(starts) - first - 40)
    t3 = time.perf_counter()
    for b in ahead(first + 40, first + 40 + n2):
        z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    hook.check()
    # (the one-stream loop again, over the batches that FOLLOW: the same stretch of the stream as the two-stream figure's neighbourhood)
    lo2 = first + 40 + n2
    n3 = min(n2, len(starts) - lo2)
    t6 = time.perf_counter()
    for i in range(lo2, lo2 + n3):
        b = loader(starts[i])
        z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
    torch.cuda.synchronize()
    t7 = time.perf_counter()
print(json.dumps({
    'edge_features': features,
    'what': 'TGAT eval forward, example dims (node 1 / edge 172 / time 100 / embed 172, 2 heads, 2 layers), 600 seeds, k=[20,20]',
    'tgat_forward_us': fwd_us, 'sampler_plus_forward_us_per_batch': 1e6 * (t2 - t0) / n, 'host_us_per_batch': 1e6 * (t1 - t0) / n,
    'sampler_plus_forward_two_streams_us_per_batch': 1e6 * (t5 - t3) / n2, 'two_streams_host_us_per_batch': 1e6 * (t4 - t3) / n2, 'two_streams_batches': n2,
    'one_stream_again_later_batches_us_per_batch': 1e6 * (t7 - t6) / max(n3, 1),
    'algorithmic_gflop_folded': 3.5, 'reference_cpu_forward_ms': 166.0,
}))

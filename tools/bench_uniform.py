#!/usr/bin/env python
"""NeighborSamplerHook (uniform temporal neighbor sampling, tgm/hooks/neighbors/uniform.py:87-142) at the headline shape: wiki-shaped stream,
seeds = src | dst | neg, num_nbrs = [20, 20], bs = 200; steady-state batches mid-stream, HIP events around the timed steps; one JSON line.
python tools/bench_uniform.py [n_steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import DGData, DGDataLoader, DGraph  # noqa: E402
from tgm_amd.hooks import HookManager, NeighborSamplerHook, RandomNegativeEdgeSamplerHook  # noqa: E402
from tgm_amd.synth import make_stream  # noqa: E402

dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
res = {}
for shape, bs, ks, validate in (('wiki', 200, [20, 20], 'sync'), ('wiki', 200, [20, 20], 'deferred'), ('review', 512, [10, 10], 'sync'),
                               ('review', 512, [10, 10], 'deferred')):
    st = make_stream(shape, seed=1337)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x, static_node_x=st.node_x), device=dev)
    hm = HookManager(keys=['k'])
    hm.register('k', RandomNegativeEdgeSamplerHook(int(st.dst.min()), st.num_nodes))
    hm.register('k', NeighborSamplerHook(ks, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], seed=7, validate=validate))
    loader = DGDataLoader(dg, batch_size=bs, hook_manager=hm)
    starts = loader._starts
    first = len(starts) // 2
    with hm.activate('k'):
        for i in range(first, first + 30):
            b = loader(starts[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(first + 30, first + 30 + n):
            b = loader(starts[i])
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        hm.active_hooks()[-1].check()
    us = e0.elapsed_time(e1) / n * 1000
    slots = 3 * bs * ks[0] + 3 * bs * ks[0] * ks[1]
    res[f'{shape}, validate={validate}'] = {'us_per_batch': us, 'host_us_per_batch': 1e6 * (t1 - t0) / n, 'sampled_edges_per_s': slots / (us * 1e-6), 'bs': bs, 'num_nbrs': ks,
                  'slots_per_batch': slots, 'D': int(st.edge_dim)}
print(json.dumps({'what': 'NeighborSamplerHook (uniform sampling over the static index, dense feature rows) behind DGDataLoader, steady-state batches mid-stream', **res}))

#!/usr/bin/env python
"""Per-step wall time of the host path, stage by stage (run on the GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import DGData, DGDataLoader, DGraph  # noqa: E402
from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook  # noqa: E402
from tgm_amd.synth import make_stream  # noqa: E402

stream = make_stream('wiki', seed=1337)
dev = torch.device('cuda', 0)
data = DGData.from_raw(stream.ts, torch.stack([stream.src, stream.dst], 1), stream.edge_x, static_node_x=stream.node_x)
dg = DGraph(data, device=dev)
keys, tkeys = ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time']


def timeit(label, hooks, n=600):
    hm = HookManager(keys=['k'])
    for h in hooks:
        hm.register('k', h)
    loader = DGDataLoader(dg, batch_size=200, hook_manager=hm if hooks else None)
    starts = loader._starts
    with hm.activate('k'):
        for i in range(20):
            loader(starts[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20, 20 + n):
            loader(starts[i])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f'{label:50s} host {1e6 * (t1 - t0) / n:7.1f} us/step   +drain {1e6 * (t2 - t1) / n:6.1f} us/step', flush=True)


neg = lambda: RandomNegativeEdgeSamplerHook(8227, stream.num_nodes)
timeit('loader only (slice + materialize)', [])
timeit('+ negatives', [neg()])
for mode in ('ring', 'csr'):
    for nb in ([20], [20, 20]):
        timeit(f'+ negatives + recency {mode} {nb} deferred', [neg(), RecencyNeighborHook(stream.num_nodes, nb, keys, tkeys, mode=mode, validate='deferred', batch_size=200)])
timeit('+ negatives + recency ring [20,20] sync-validate', [neg(), RecencyNeighborHook(stream.num_nodes, [20, 20], keys, tkeys)])
timeit('+ negatives + recency ring [20,20] off', [neg(), RecencyNeighborHook(stream.num_nodes, [20, 20], keys, tkeys, validate='off')])

# raw cost of the step call alone (same argument block re-submitted)
from tgm_amd import _native
hook = RecencyNeighborHook(stream.num_nodes, [20, 20], keys, tkeys, validate='deferred')
hm = HookManager(keys=['k']); hm.register('k', neg()); hm.register('k', hook)
loader = DGDataLoader(dg, batch_size=200, hook_manager=hm)
with hm.activate('k'):
    b = loader(loader._starts[5])
lib = _native.load()
st = hook._step
sp = _native.stream_ptr(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    lib.tgmx_recency_step(st, sp)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{"tgmx_recency_step alone":50s} host {1e6 * (t1 - t0) / 300:7.1f} us/call   total {1e6 * (t2 - t0) / 300:7.1f} us/call', flush=True)

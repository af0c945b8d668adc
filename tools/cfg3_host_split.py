#!/usr/bin/env python
"""cfg 3: where the consumer thread's time goes per batch (perf_counter around next(loader), TGNStep.batch and the native calls inside them)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import DGData, DGDataLoader, DGraph, _native
from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook, SampledEdgeListHook
from tgm_amd.nn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, TGNMemory, TGNStep
from tgm_amd.synth import make_stream

dev = torch.device('cuda', 0)
st = make_stream('review', seed=1337, device=dev)
N, D, M, T_, bs, ks = st.num_nodes, st.edge_dim, 100, 100, 512, [10, 10]
dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=dev)
hm = HookManager(keys=['k'])
hm.register('k', RandomNegativeEdgeSamplerHook(int(st.dst.min()), N))
hm.register('k', RecencyNeighborHook(N, ks, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], validate='deferred', edge_features='by_id'))
hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
hm.register('k', SampledEdgeListHook(hop=0))
mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator()).to(dev).train()
mem.reuse_forward = True
enc = GraphAttentionEmbedding(M, 100, D, mem.time_enc).to(dev).eval()
step = TGNStep(mem, enc)
lib = _native.load()
acc = {}


def timed(name):
    f = getattr(lib, name)

    def w(*a):
        t = time.perf_counter()
        r = f(*a)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    setattr(lib, name, w)


for nm in ('tgmx_tgn_step', 'tgmx_worker_pipeline_step', 'tgmx_worker_wait', 'tgmx_event_synchronize', 'tgmx_event_record', 'tgmx_stream_wait_event'):
    timed(nm)


def batches(lo, hi):
    return DGDataLoader(dg.slice_events(lo * bs, hi * bs), batch_size=bs, hook_manager=hm, output_pool=3, prefetch=2, side_stream=True)


n = 400
with hm.activate('k'), torch.no_grad():
    for b in batches(0, 300):
        step.batch(b)
    torch.cuda.synchronize()
    acc.clear()
    t_next = t_step = 0.0
    it = iter(batches(300, 300 + n))
    t0 = time.perf_counter()
    while True:
        a = time.perf_counter()
        try:
            b = next(it)
        except StopIteration:
            break
        c = time.perf_counter()
        step.batch(b)
        d = time.perf_counter()
        t_next += c - a
        t_step += d - c
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
us = lambda x: round(1e6 * x / n, 1)
print(json.dumps({'us_per_batch': us(t2 - t0), 'issue_us_per_batch': us(t1 - t0), 'next(loader)': us(t_next), 'step.batch': us(t_step),
                  'native': {k: us(v) for k, v in sorted(acc.items())}}))

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


if os.environ.get('SPLIT_QUERY'):  # is the batch's sizes event complete when the consumer asks for it?  (hipEventQuery in front of the synchronize)
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipEventQuery.argtypes = [ctypes.c_void_p]
    _sync0 = lib.tgmx_event_synchronize
    q = {'ready': 0, 'not_ready': 0, 'query_s': 0.0, 'sync_ready_s': 0.0, 'sync_not_ready_s': 0.0}

    def _sync(ev):
        t = time.perf_counter()
        r = hip.hipEventQuery(ev)
        t1 = time.perf_counter()
        rc = _sync0(ev)
        t2 = time.perf_counter()
        q['query_s'] += t1 - t
        if r == 0:
            q['ready'] += 1
            q['sync_ready_s'] += t2 - t1
        else:
            q['not_ready'] += 1
            q['sync_not_ready_s'] += t2 - t1
        return rc
    lib.tgmx_event_synchronize = _sync
    acc['query'] = q
for nm in ('tgmx_tgn_step', 'tgmx_worker_pipeline_step', 'tgmx_worker_wait', 'tgmx_event_synchronize', 'tgmx_event_record', 'tgmx_stream_wait_event'):
    timed(nm)


def batches(lo, hi):
    return DGDataLoader(dg.slice_events(lo * bs, hi * bs), batch_size=bs, hook_manager=hm, output_pool=3, prefetch=2, side_stream=True)


n = 400
LOADER_EV = bool(os.environ.get('SPLIT_LOADER_EVENTS'))  # with TGMX_LOADER_WORKER=0: device-side duration of every production on the loader's stream
lev = []
if LOADER_EV:
    _call0 = DGDataLoader.__call__

    def _call(self, *a, **kw):
        if not kw.get('_deferred'):
            return _call0(self, *a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()  # (the current stream is the loader's own here)
        out = _call0(self, *a, **kw)
        e1.record()
        lev.append((e0, e1))
        return out
    DGDataLoader.__call__ = _call
EV = bool(os.environ.get('SPLIT_EVENTS'))  # device-side start / end of every model step on the caller's stream (two more event records per batch)
evs = []
with hm.activate('k'), torch.no_grad():
    for b in batches(0, 300):
        step.batch(b)
    torch.cuda.synchronize()
    qq = acc.get('query')
    acc.clear()
    if qq is not None:
        for k_ in qq:
            qq[k_] = 0
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
        if EV:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        step.batch(b)
        if EV:
            e1.record()
            evs.append((e0, e1))
        d = time.perf_counter()
        t_next += c - a
        t_step += d - c
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
us = lambda x: round(1e6 * x / n, 1)
loader_side = None
if LOADER_EV and lev:
    dur = sorted(1e3 * a.elapsed_time(b_) for a, b_ in lev[-300:])
    gap = sorted(1e3 * lev[i][1].elapsed_time(lev[i + 1][0]) for i in range(len(lev) - 300, len(lev) - 1))
    loader_side = {'production_start_to_end_on_device_us_median': round(dur[len(dur) // 2], 1), 'p90': round(dur[int(len(dur) * 0.9)], 1),
                   'loader_stream_idle_between_productions_us_median': round(gap[len(gap) // 2], 1)}
dev_side = None
if EV:
    chain = [1e3 * a.elapsed_time(b_) for a, b_ in evs[50:]]
    idle = [1e3 * evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(50, len(evs) - 1)]
    chain.sort(); idle.sort()
    dev_side = {'model_step_start_to_end_on_device_us_median': round(chain[len(chain) // 2], 1), 'p90': round(chain[int(len(chain) * 0.9)], 1),
                'caller_stream_idle_between_steps_us_median': round(idle[len(idle) // 2], 1), 'idle_p90': round(idle[int(len(idle) * 0.9)], 1)}
print(json.dumps({'us_per_batch': us(t2 - t0), 'issue_us_per_batch': us(t1 - t0), 'next(loader)': us(t_next), 'step.batch': us(t_step),
                  'native': {k: us(v) for k, v in sorted(acc.items()) if k != 'query'}, 'device': dev_side, 'sizes_event': qq, 'loader_stream': loader_side}))

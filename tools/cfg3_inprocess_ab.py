#!/usr/bin/env python
"""cfg 3 (TGN step + prefetching loader) A/B INSIDE one process: the per-call knobs of tgmx_tgn_step / the loader are switched between segments of
the same run (box-to-box and run-to-run drift of the host side is 10-20 % on this pool -- larger than the effects under test).
usage: cfg3_inprocess_ab.py ROUNDS SEG "A=1 B=0" "C=1" ...   (each argument one configuration = environment assignments; "-" = defaults)"""
import contextlib, json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import DGData, DGDataLoader, DGraph
from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook, SampledEdgeListHook
from tgm_amd.nn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, TGNMemory, TGNStep
from tgm_amd.synth import make_stream

rounds, seg = int(sys.argv[1]), int(sys.argv[2])
cfgs = sys.argv[3:]
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
keys = sorted({kv.split('=')[0] for c in cfgs for kv in c.split() if '=' in kv})


def batches(lo, hi):
    pf = int(os.environ.get('AB_PREFETCH', '2'))  # (the loader's own arguments, switchable per segment like the library's knobs; 3 and 4 measured no better than 2)
    return DGDataLoader(dg.slice_events(lo * bs, hi * bs), batch_size=bs, hook_manager=hm, output_pool=int(os.environ.get('AB_POOL', str(pf + 1))), prefetch=pf, side_stream=True)


res = {c: [] for c in cfgs}
high = torch.cuda.Stream(device=dev, priority=-1)
with hm.activate('k'), torch.no_grad():
    for b in batches(0, 200):
        step.batch(b)
    torch.cuda.synchronize()
    lead = 300
    for r in range(rounds):
        order = cfgs if r % 2 == 0 else cfgs[::-1]
        for c in order:
            for k in keys:
                os.environ.pop(k, None)
            for kv in c.split():
                if '=' in kv:
                    k, v = kv.split('=')
                    os.environ[k] = v
            # every segment replays the SAME stretch of the stream from a reset state (identical work per segment): `lead` batches untimed, `seg` timed
            hm.reset_state()
            mem.reset_state()
            torch.cuda.synchronize()
            # AB_MAIN_HIGH=1: the consumer's (model's) stream is a high-priority one of its own instead of the default stream
            ctx = torch.cuda.stream(high) if os.environ.get('AB_MAIN_HIGH') == '1' else contextlib.nullcontext()
            with ctx:
                for b in batches(0, lead):
                    step.batch(b)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for b in batches(lead, lead + seg):
                    step.batch(b)
                torch.cuda.synchronize()
                res[c].append(1e6 * (time.perf_counter() - t0) / seg)
for c in cfgs:
    v = res[c]
    print(json.dumps({'config': c, 'median_us_per_batch': round(statistics.median(v), 1), 'min': round(min(v), 1), 'all': [round(x, 1) for x in v]}))

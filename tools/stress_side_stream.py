#!/usr/bin/env python
"""Shake the timing of DGDataLoader(side_stream=True): random device-side delays on the caller's stream and on the loader's stream, random host
sleeps, several passes -- every pass must equal the single-stream reference bit for bit.  (A race in the event protocol shows up as a mismatch.)"""
import os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import DGData, DGDataLoader, DGraph
from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook, SampledEdgeListHook
from tgm_amd.nn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, TGNMemory, TGNStep
from tgm_amd.synth import make_stream

DEV = 'cuda'
features = sys.argv[1] if len(sys.argv) > 1 else 'by_id'
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
use_step = len(sys.argv) > 3 and sys.argv[3] == 'step'  # the model side through TGNStep (tgmx_tgn_step: its own fork / join inside) in the side-stream passes
st = make_stream('review', seed=8, num_edges=20 * 512 + 77, n_src=3000, n_dst=400)
N, D, M, T_, bs = st.num_nodes, 16, 100, 100, 512


def run(side, jitter, seed):
    rnd = random.Random(seed)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', RandomNegativeEdgeSamplerHook(3000, N, seed=4))
    hook = RecencyNeighborHook(N, [10, 10], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], validate='deferred', edge_features=features)
    hm.register('k', hook)
    hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
    hm.register('k', SampledEdgeListHook(hop=0))
    torch.manual_seed(0)
    mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator()).to(DEV).train()
    mem.reuse_forward = True
    enc = GraphAttentionEmbedding(M, 100, D, mem.time_enc).to(DEV).eval()
    step = TGNStep(mem, enc) if (use_step and side) else None
    out = []
    with hm.activate('k'), torch.no_grad():
        for ep in range(2):
            hm.reset_state()
            ld = DGDataLoader(dg, batch_size=bs, hook_manager=hm, output_pool=2, prefetch=1, side_stream=side)
            for batch in ld:
                if jitter:
                    r = rnd.random()
                    if r < 0.3:
                        torch.cuda._sleep(int(rnd.random() * 2_000_000))  # up to ~1 ms on the caller's stream
                    elif r < 0.5 and ld._side is not None:
                        with torch.cuda.stream(ld._side[0]):
                            torch.cuda._sleep(int(rnd.random() * 2_000_000))  # ... on the loader's stream
                    elif r < 0.6:
                        time.sleep(rnd.random() * 0.002)
                if step is not None:
                    z2, z, lu = step.batch(batch)
                else:
                    z, lu = mem(batch.unique_nids)
                    z2 = enc(z, lu, batch.sampled_edge_index, batch.sampled_edge_time, batch.sampled_edge_x)
                    mem.update_state(batch.edge_src, batch.edge_dst, batch.edge_time, batch.edge_x)
                out.append([t.clone() for t in (batch.neg, batch.unique_nids, batch.nbr_nids[0], batch.nbr_nids[1], batch.nbr_edge_time[1],
                                                batch.sampled_edge_index, batch.sampled_edge_time, batch.sampled_edge_x, z, lu, z2)])
        hook.check(); mem.check()
    torch.cuda.synchronize()
    return out, mem.memory.clone()


ref, mref = run(False, False, 0)
bad = 0
for p in range(passes):
    got, m = run(True, True, p + 1)
    for b, (x, y) in enumerate(zip(ref, got)):
        for i, (u, v) in enumerate(zip(x, y)):
            if not torch.equal(u, v):
                bad += 1
                print(f'pass {p}: batch {b} item {i} differs ({(u != v).sum().item()} elements)', flush=True)
                break
    if not torch.equal(m, mref):
        print(f'pass {p}: final memory differs', flush=True)
print('mismatches:', bad)

import cProfile, pstats, os, sys, time
import torch
sys.path.insert(0, '/root/repo')
from tgm_amd import DGData, DGDataLoader, DGraph
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
mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator()).to(dev).train(); mem.reuse_forward = True
enc = GraphAttentionEmbedding(M, 100, D, mem.time_enc).to(dev).eval()
step = TGNStep(mem, enc)
def batches(lo, hi):
    return DGDataLoader(dg.slice_events(lo * bs, hi * bs), batch_size=bs, hook_manager=hm, output_pool=3, prefetch=2, side_stream=True)
with hm.activate('k'), torch.no_grad():
    for b in batches(0, 100): step.batch(b)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for b in batches(100, 500): step.batch(b)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(30)

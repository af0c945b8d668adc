"""End-to-end wiring of the BASELINE.json configurations on the GPU (small sizes), each checked
against the CPU oracles: cfg 1 (1-hop k=10 sampler), cfg 3 (TGN: sampler + dedup + memory +
graph attention, the loop of examples/linkproppred/tgn.py), cfg 5 (yearly snapshots -> TGCN)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def close(got, ref, tag, rtol=1e-5):
    err = (got - ref).abs()
    worst = (err / (rtol * ref.abs().clamp(min=1.0))).max().item() if ref.numel() else 0.0
    assert worst <= 1.0, f'{tag}: {worst:.2f}x the bound'


def test_cfg1_wiki_one_hop_k10():
    from oracle.ring_port import RingSamplerCPU
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RecencyNeighborHook
    from tgm_amd.synth import make_stream

    st = make_stream('wiki', seed=2, num_edges=8000, edge_dim=172)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', RecencyNeighborHook(st.num_nodes, [10], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time']))
    ref = RingSamplerCPU(st.num_nodes, [10], 172)
    with hm.activate('k'):
        for b, batch in enumerate(DGDataLoader(dg, batch_size=200, hook_manager=hm)):
            lo, hi = b * 200, min((b + 1) * 200, st.num_edges)
            (_, _, n, t, x), = ref.step(torch.cat([st.src[lo:hi], st.dst[lo:hi]]), torch.cat([st.ts[lo:hi]] * 2), st.src[lo:hi], st.dst[lo:hi],
                                       st.ts[lo:hi], st.edge_x[lo:hi])  # fmt: skip
            assert torch.equal(batch.nbr_nids[0].cpu(), n) and torch.equal(batch.nbr_edge_time[0].cpu(), t) and torch.equal(batch.nbr_edge_x[0].cpu(), x)


def test_cfg3_tgn_pipeline():
    """Per batch: negatives -> recency sampler [10] -> dedup -> memory(unique ids) -> graph attention embedding ->
    update_state, exactly the order of examples/linkproppred/tgn.py:71-118 (train-mode memory, eval-mode conv)."""
    from oracle.ring_port import RingSamplerCPU
    from oracle.tgn_ref import TGNMemoryRef, graph_attention_embedding_ref
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook
    from tgm_amd.nn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, TGNMemory
    from tgm_amd.synth import make_stream

    st = make_stream('review', seed=8, num_edges=4096, n_src=700, n_dst=100)
    ts = st.ts[0] + torch.arange(st.num_edges) * 300  # no float32 time ties inside a batch
    N, D, M, T_, bs, k = st.num_nodes, 16, 100, 100, 512, 10
    dg = DGraph(DGData.from_raw(ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', RandomNegativeEdgeSamplerHook(700, N, seed=4))
    hm.register('k', RecencyNeighborHook(N, [k], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time']))
    hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
    torch.manual_seed(0)
    mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator()).to(DEV).train()
    enc = GraphAttentionEmbedding(M, 100, D, mem.time_enc).to(DEV).eval()
    mp = {k_: v.detach().cpu().clone() for k_, v in mem.state_dict().items() if k_ not in ('memory', 'last_update', '_assoc')}
    ep = {k_: v.detach().cpu().clone() for k_, v in enc.state_dict().items()}
    ref_mem, ref_ring = TGNMemoryRef(N, D, M, T_, mp, 'last'), RingSamplerCPU(N, [k], D)
    with hm.activate('k'):
        for b, batch in enumerate(DGDataLoader(dg, batch_size=bs, hook_manager=hm)):
            lo, hi = b * bs, min((b + 1) * bs, st.num_edges)
            nbr = batch.nbr_nids[0].flatten()
            keep = nbr != -1
            seeds = torch.cat([batch.edge_src, batch.edge_dst, batch.neg]).repeat_interleave(k)
            edge_index = torch.stack([batch.global_to_local(seeds[keep]), batch.global_to_local(nbr[keep])]).long()
            e_t = batch.nbr_edge_time[0].flatten()[keep]
            e_x = batch.nbr_edge_x[0].flatten(0, -2)[keep]
            z, lu = mem(batch.unique_nids)
            z2 = enc(z, lu, edge_index, e_t, e_x)
            mem.update_state(batch.edge_src, batch.edge_dst, batch.edge_time, batch.edge_x)
            # ---- oracle side, same inputs ----
            neg = batch.neg.cpu()
            (_, _, n_ref, t_ref, x_ref), = ref_ring.step(torch.cat([st.src[lo:hi], st.dst[lo:hi], neg]), torch.cat([ts[lo:hi]] * 3), st.src[lo:hi],
                                                       st.dst[lo:hi], ts[lo:hi], st.edge_x[lo:hi])  # fmt: skip
            assert torch.equal(batch.nbr_nids[0].cpu(), n_ref)
            uniq = torch.unique(torch.cat([st.src[lo:hi], st.dst[lo:hi], neg, n_ref.flatten()[n_ref.flatten() != -1]]))
            assert torch.equal(batch.unique_nids.cpu(), uniq)
            zr, lur = ref_mem.forward(uniq.long())
            close(z.cpu(), zr, f'b{b} memory')
            assert torch.equal(lu.cpu(), lur)
            close(z2.cpu(), graph_attention_embedding_ref(ep, zr, lur, edge_index.cpu(), e_t.cpu(), e_x.cpu()), f'b{b} embedding')
            ref_mem.update_state(st.src[lo:hi], st.dst[lo:hi], ts[lo:hi], st.edge_x[lo:hi])
    close(mem.memory.cpu(), ref_mem.memory, 'final memory')


@pytest.mark.parametrize('features', ['by_id', 'dense'])
def test_side_stream_loader_gives_the_single_stream_results(features):
    """DGDataLoader(side_stream=True): the hooks' chain of batch i + 1 runs on the loader's own stream beside the model's chain of
    batch i, ordered by events only.  Every batch attribute and every model output must equal the single-stream pass's bit for bit
    (TGN loop: sampler -> dedup -> edge list -> memory -> embedding -> update_state, two passes over the stream with a reset between),
    including what a racy ordering would break: the sampler's rings, the recycled output sets (pool of two), the final memory."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook, SampledEdgeListHook
    from tgm_amd.nn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, TGNMemory
    from tgm_amd.synth import make_stream

    st = make_stream('review', seed=8, num_edges=20 * 512 + 77, n_src=3000, n_dst=400)
    N, D, M, T_, bs = st.num_nodes, 16, 100, 100, 512

    def run(side):
        dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
        hm = HookManager(keys=['k'])
        hm.register('k', RandomNegativeEdgeSamplerHook(3000, N, seed=4))
        hook = RecencyNeighborHook(N, [10, 10], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], validate='deferred',
                                   edge_features=features)
        hm.register('k', hook)
        hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
        hm.register('k', SampledEdgeListHook(hop=0))
        torch.manual_seed(0)
        mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator()).to(DEV).train()
        mem.reuse_forward = True
        enc = GraphAttentionEmbedding(M, 100, D, mem.time_enc).to(DEV).eval()
        out = []
        with hm.activate('k'), torch.no_grad():
            for ep in range(2):
                hm.reset_state()
                for batch in DGDataLoader(dg, batch_size=bs, hook_manager=hm, output_pool=2, prefetch=1, side_stream=side):
                    z, lu = mem(batch.unique_nids)
                    z2 = enc(z, lu, batch.sampled_edge_index, batch.sampled_edge_time, batch.sampled_edge_x)
                    mem.update_state(batch.edge_src, batch.edge_dst, batch.edge_time, batch.edge_x)
                    out.append([t.clone() for t in (batch.neg, batch.unique_nids, batch.nbr_nids[0], batch.nbr_nids[1], batch.nbr_edge_time[1],
                                                    batch.sampled_edge_index, batch.sampled_edge_time, batch.sampled_edge_x, z, lu, z2)])
            hook.check()
            mem.check()
        torch.cuda.synchronize()
        return out, mem.memory.clone(), mem.last_update.clone()

    one, mem1, lu1 = run(False)
    two, mem2, lu2 = run(True)
    assert len(one) == len(two) == 2 * 21
    for b, (x, y) in enumerate(zip(one, two)):
        for i, (u, v) in enumerate(zip(x, y)):
            assert torch.equal(u, v), f'batch {b} item {i}'
    assert torch.equal(mem1, mem2) and torch.equal(lu1, lu2)


def test_two_side_stream_loaders_interleaved_share_one_stream():
    """Every DGDataLoader(side_stream=True) of a process issues on ONE loader stream per device (a stream per loader made the second, third, ...
    loader land on a hardware queue the caller's or the library's streams use: cfg 3 ran at the one-stream speed through every loader but the
    first).  Two loaders over two graphs with their own hooks, consumed alternately (a train / validation interleaving): every batch equals
    the single-stream pass's bit for bit, and the two loaders hold the same stream object."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook, SampledEdgeListHook
    from tgm_amd.synth import make_stream

    def pipeline(seed, side):
        st = make_stream('review', seed=seed, num_edges=12 * 256 + 31, n_src=900, n_dst=150)
        dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
        hm = HookManager(keys=['k'])
        hm.register('k', RandomNegativeEdgeSamplerHook(900, st.num_nodes, seed=seed))
        hook = RecencyNeighborHook(st.num_nodes, [10, 10], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], validate='deferred',
                                   edge_features='by_id')
        hm.register('k', hook)
        hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
        hm.register('k', SampledEdgeListHook(hop=0))
        kw = dict(output_pool=3, prefetch=2, side_stream=True) if side else {}
        return hm, hook, DGDataLoader(dg, batch_size=256, hook_manager=hm, **kw)

    def grab(batch):
        return [t.clone() for t in (batch.neg, batch.unique_nids, batch.nbr_nids[0], batch.nbr_nids[1], batch.nbr_edge_time[1], batch.sampled_edge_index,
                                    batch.sampled_edge_time, batch.sampled_edge_x)]

    def run(side):
        (hm_a, hook_a, ld_a), (hm_b, hook_b, ld_b) = pipeline(21, side), pipeline(22, side)
        out_a, out_b = [], []
        # a HookManager activation is a context of its own: the two pipelines alternate batch by batch, each under its own manager
        it_a = it_b = None
        for _ in range(13):
            with hm_a.activate('k'):
                it_a = it_a or iter(ld_a)
                out_a.append(grab(next(it_a)))
            with hm_b.activate('k'):
                it_b = it_b or iter(ld_b)
                out_b.append(grab(next(it_b)))
        for it in (it_a, it_b):
            it.close()  # (the generators hand their stream's work back to the caller's stream)
        hook_a.check()
        hook_b.check()
        torch.cuda.synchronize()
        return out_a, out_b, ld_a, ld_b

    a1, b1, _, _ = run(False)
    a2, b2, ld_a, ld_b = run(True)
    assert ld_a._side[0] is ld_b._side[0]
    for name, one, two in (('a', a1, a2), ('b', b1, b2)):
        assert len(one) == len(two) == 13
        for i, (x, y) in enumerate(zip(one, two)):
            for j, (u, v) in enumerate(zip(x, y)):
                assert torch.equal(u, v), f'loader {name} batch {i} item {j}'


@pytest.mark.parametrize('validate', ['deferred', 'sync'])
def test_side_stream_with_timed_steps_keeps_the_batch_order(validate):
    """A TIMED step (``profile_hop``: HIP events around the dominant launch) is issued from the calling thread while earlier batches' steps
    may still sit in the launch worker's queue: it must land on the loader's stream behind them (``CompiledPipeline._order_behind_worker``),
    or batch j's ring update overtakes batch j - 1's lookups.  Every second step timed, delays on the loader's stream to keep the worker's
    queue full; sampler outputs equal to the single-stream pass bit for bit.  ``validate='sync'``: the status read waits for the worker too."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd._native import KernelTimer
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook
    from tgm_amd.synth import make_stream

    st = make_stream('review', seed=9, num_edges=40 * 256 + 31, n_src=2000, n_dst=300)
    N, bs = st.num_nodes, 256

    def run(side):
        dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
        hm = HookManager(keys=['k'])
        hm.register('k', RandomNegativeEdgeSamplerHook(2000, N, seed=4))
        hook = RecencyNeighborHook(N, [10, 10], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], validate=validate)
        hm.register('k', hook)
        hook.profile_hop, hook.profile_every, hook.profile_log = 1, 2, []
        hook.profile_pool = [KernelTimer() for _ in range(32)]
        out = []
        with hm.activate('k'):
            kw = dict(output_pool=3, prefetch=2, side_stream=True) if side else dict(output_pool=3, prefetch=2)
            for i, batch in enumerate(DGDataLoader(dg, batch_size=bs, hook_manager=hm, **kw)):
                if side and i % 3 == 0:
                    torch.cuda._sleep(2_000_000)  # the consumer's stream lags: productions (and the worker's queue) pile up
                out.append([t.clone() for t in (batch.neg, batch.nbr_nids[0], batch.nbr_nids[1], batch.nbr_edge_time[1], batch.nbr_edge_x[1])])
            hook.check()
        torch.cuda.synchronize()
        assert len(hook.profile_log) >= 16
        return out

    one, two = run(False), run(True)
    assert len(one) == len(two) == 41
    for b, (x, y) in enumerate(zip(one, two)):
        for i, (u, v) in enumerate(zip(x, y)):
            assert torch.equal(u, v), f'batch {b} item {i}'


def test_side_stream_argument_checks():
    from tgm_amd import DGData, DGDataLoader, DGraph

    ts = torch.arange(40)
    dg = DGraph(DGData.from_raw(ts, torch.randint(0, 5, (40, 2), dtype=torch.int32)), device=DEV)
    with pytest.raises(ValueError, match='side_stream'):
        DGDataLoader(dg, batch_size=4, side_stream=True)
    with pytest.raises(ValueError, match='output_pool'):
        DGDataLoader(dg, batch_size=4, side_stream=True, prefetch=1, output_pool=1)
    with pytest.raises(ValueError, match='side_stream'):
        DGDataLoader(dg, batch_size=4, side_stream=True, prefetch=1)  # fresh-tensor sets (output_pool=None) are not a recycled pool
    assert len(list(DGDataLoader(dg, batch_size=4, side_stream=True, prefetch=1, output_pool=2))) == 10  # no hooks: batches are views
    it = iter(DGDataLoader(dg, batch_size=4, side_stream=True, prefetch=1, output_pool=2))  # a consumer that stops early
    next(it), next(it)
    it.close()


def test_cfg5_snapshots_tgcn():
    """tgbn-trade-like: seconds -> discretize to years -> one snapshot per batch -> TGCN with carried state."""
    from oracle.tgcn_ref import tgcn_cell_ref
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.nn import TGCN

    rng = np.random.default_rng(0)
    N, E, year = 255, 30_000, 365 * 24 * 3600
    ts = torch.from_numpy(np.sort(rng.integers(0, 12 * year, E)))
    ei = torch.from_numpy(rng.integers(0, N, (E, 2)).astype(np.int32))
    data = DGData.from_raw(ts, ei, torch.rand(E, 1), static_node_x=torch.randn(N, 16), time_delta='s').discretize('Y', device=DEV)
    assert data.time_delta.unit == 'Y' and int(data.time.max()) == 11
    dg = DGraph(data, device=DEV)
    torch.manual_seed(0)
    cell = TGCN(16, 32).to(DEV).eval()
    params = {k: v.detach().cpu() for k, v in cell.state_dict().items()}
    H = H_ref = None
    n_snap = 0
    for batch in DGDataLoader(dg, batch_unit='Y'):
        edge_index = torch.stack([batch.edge_src, batch.edge_dst])
        H = cell(dg.static_node_x, edge_index, None, H)
        H_ref = tgcn_cell_ref(params, data.static_node_x, edge_index.cpu(), None, H_ref)
        close(H.cpu(), H_ref, f'snapshot {n_snap}')
        n_snap += 1
    assert n_snap == 12


def test_dedup_hook_matches_reference_golden():
    """recency sampler k=[3,2] + DeduplicationHook on the device against the reference's recorded unique_nids /
    global_to_local (golden g7)."""
    import json
    import os

    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import DeduplicationHook, HookManager, RecencyNeighborHook
    from tgm_amd.hooks.base import StatelessHook

    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g7_dedup.npz'))
    meta = json.loads(bytes(z['meta']).decode())
    neg = torch.from_numpy(z['neg']).to(DEV)

    class Replay(StatelessHook):
        _cls_requires = {'edge_src', 'edge_dst', 'edge_time'}
        _cls_produces = {'neg', 'neg_time'}

        def __init__(self):
            super().__init__()
            self.__post_init__()

        def __call__(self, dg, batch):
            batch.neg = neg[batch._edge_lo : batch._edge_lo + batch.edge_src.numel()].clone()
            batch.neg_time = batch.edge_time.clone()
            return batch

    dg = DGraph(DGData.from_raw(torch.from_numpy(z['ts']), torch.stack([torch.from_numpy(z['src']), torch.from_numpy(z['dst'])], 1)), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', Replay())
    hm.register('k', RecencyNeighborHook(meta['num_nodes'], meta['num_nbrs'], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time']))
    hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
    nb = 0
    with hm.activate('k'):
        for b, batch in enumerate(DGDataLoader(dg, batch_size=meta['batch_size'], hook_manager=hm)):
            assert batch.unique_nids.dtype == torch.int32
            assert np.array_equal(batch.unique_nids.cpu().numpy(), z[f'b{b}_unique_nids']), f'batch {b}'
            assert np.array_equal(batch.global_to_local(batch.edge_src).cpu().numpy(), z[f'b{b}_local_src']), f'batch {b}'
            nb += 1
    assert nb == meta['num_batches']


def test_random_negatives_on_device():
    """One launch for neg + neg_time: range, dtype, a fresh copy of the times, seed-reproducible, roughly uniform."""
    from tgm_amd import DGData, DGraph
    from tgm_amd.hooks import RandomNegativeEdgeSamplerHook

    E = 50_000
    ts = torch.arange(E, dtype=torch.int64)
    ei = torch.stack([torch.zeros(E, dtype=torch.int32), torch.ones(E, dtype=torch.int32)], 1)
    dg = DGraph(DGData.from_raw(ts, ei), device=DEV)
    batch = dg.slice_events(0, E).materialize()
    outs = []
    for rep in range(2):
        h = RandomNegativeEdgeSamplerHook(100, 164, seed=11)
        b1 = h(dg, dg.slice_events(0, E).materialize())
        b2 = h(dg, dg.slice_events(0, E).materialize())
        outs.append((b1.neg.cpu(), b2.neg.cpu()))
        assert b1.neg.dtype == torch.int32 and b1.neg.shape == (E,) and int(b1.neg.min()) >= 100 and int(b1.neg.max()) < 164
        assert torch.equal(b1.neg_time, batch.edge_time) and b1.neg_time.data_ptr() != b1.edge_time.data_ptr()
        assert not torch.equal(b1.neg, b2.neg)  # consecutive calls draw different values
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])  # same seed, same call order
    counts = torch.bincount(outs[0][0] - 100, minlength=64).double()
    sigma = (E / 64 * (1 - 1 / 64)) ** 0.5
    assert bool(((counts - E / 64).abs() < 5 * sigma).all()), counts


@pytest.mark.parametrize('N,n_ids', [(31, 100), (1024 * 32, 5000), (200_003, 170_000), (1_000_000, 600_000)])
def test_unique_ids_matches_torch_unique(N, n_ids):
    """tgmx_unique_ids (bitmap + multi-round popcount scan) against torch.unique, with pads and several parts."""
    import ctypes

    from tgm_amd import _native

    lib = _native.load()
    g = torch.Generator().manual_seed(N)
    parts = [torch.randint(0, N, (n_ids // 3,), generator=g, dtype=torch.int32), torch.randint(0, max(N // 7, 1), (n_ids // 3,), generator=g, dtype=torch.int32),
             torch.randint(-1, N, (n_ids - 2 * (n_ids // 3),), generator=g, dtype=torch.int32), torch.tensor([N - 1, 0, -1], dtype=torch.int32)]  # fmt: skip
    dev_parts = [p.to(DEV) for p in parts]
    allv = torch.cat(parts)
    exp = torch.unique(allv[allv >= 0])
    ws = torch.zeros(int(lib.tgmx_unique_ids_workspace_bytes(N)), dtype=torch.uint8, device=DEV)  # zero at first use; every call leaves it zero
    out = torch.empty(min(allv.numel(), N), dtype=torch.int32, device=DEV)
    cs = torch.zeros(2, dtype=torch.int64, device=DEV)
    ptrs = (ctypes.c_void_p * 4)(*[p.data_ptr() for p in dev_parts])
    sizes = (ctypes.c_int64 * 4)(*[p.numel() for p in dev_parts])
    rc = lib.tgmx_unique_ids(ptrs, sizes, 4, N, ws.data_ptr(), out.data_ptr(), cs[0:1].data_ptr(), cs[1:2].view(torch.int32)[0:1].data_ptr(),
                             _native.stream_ptr(0))
    assert rc == 0
    cnt, st = cs.tolist()
    assert st == 0 and cnt == exp.numel()
    assert torch.equal(out[:cnt].cpu(), exp)
    assert not bool(ws.any()), 'the bitmap must be clean again after the call'

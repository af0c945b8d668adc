"""GPU parity of the TGN memory path: HIP TGNMemory vs the reference's outputs (goldens g8) and the
oracle; GraphAttentionEmbedding vs our CPU restatement of the published TransformerConv definition
(third-party arithmetic: unpinned upstream).  Tolerance 1e-5 relative (floor 1e-5), ints exact."""
import pytest
import torch

import golden_util as gu
from test_tgn_oracle_cpu import CASES, close, drive

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _memory(meta, a):
    from tgm_amd.nn import IdentityMessage, LastAggregator, MeanAggregator, TGNMemory

    D, M, T = meta['raw_msg_dim'], meta['memory_dim'], meta['time_dim']
    aggr = LastAggregator() if meta['aggr'] == 'last' else MeanAggregator()
    mem = TGNMemory(meta['num_nodes'], D, M, T, message_module=IdentityMessage(D, M, T), aggregator_module=aggr).to(DEV)
    missing = mem.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in a.items() if k.startswith('w_')}, strict=False)
    assert set(missing.missing_keys) <= {'memory', 'last_update', '_assoc'} and not missing.unexpected_keys
    mem.reset_state()
    return mem.train()


def _grad_mode(inference):
    """The modules dispatch on the autograd state: grad-enabled calls take the saving forward (autograd Functions), no_grad calls
    the one-call inference drivers.  Every oracle comparison below runs both."""
    return torch.no_grad() if inference else torch.enable_grad()


@pytest.mark.parametrize('inference', [False, True])
@pytest.mark.parametrize('case', CASES)
def test_tgn_memory_matches_reference(case, inference):
    meta, a = gu.load(case)
    T = torch.from_numpy
    mem = _memory(meta, a)
    with _grad_mode(inference):
        _check_memory_golden(case, meta, a, mem, T)


def _check_memory_golden(case, meta, a, mem, T):
    for ev in drive(meta, a, mem, to=lambda t: t.to(DEV)):
        if ev[0] == 'fwd':
            _, b, z, lu = ev
            close(z.cpu(), T(a[f'b{b}_z']), f'{case} b{b} z')
            assert torch.equal(lu.cpu(), T(a[f'b{b}_last_update'])), f'{case} b{b} last_update'
        elif ev[0] == 'state':
            b = ev[1]
            close(mem.memory.cpu(), T(a[f'b{b}_memory']), f'{case} b{b} memory')
            assert torch.equal(mem.last_update.cpu(), T(a[f'b{b}_mem_last_update']))
        else:
            close(mem.memory.cpu(), T(a['flush_memory']), f'{case} flush')
            assert torch.equal(mem.last_update.cpu(), T(a['flush_last_update']))


@pytest.mark.parametrize('aggr,bs,log_cap,inference', [('last', 512, None, False), ('last', 512, None, True), ('mean', 512, None, False), ('mean', 512, None, True),
                                                       ('last', 700, None, True), ('mean', 100, None, False), ('last', 4096, None, False), ('last', 4096, None, True),
                                                       ('last', 512, 64, False), ('mean', 100, 256, True)])
def test_tgn_memory_matches_oracle_review_shaped(aggr, bs, log_cap, inference):
    """Example dims (memory/time 100, msg 16) on a review-shaped stream with hubs; bs=512 is the BASELINE batch (one-launch
    id grouping: 2*bs <= 1024), bs=700 and bs=4096 (BASELINE cfg 4's batch) take tgmx_group_ids_large for the message store / commit.  log_cap: a tiny
    message-log capacity, so the store is compacted every few batches (the log must not grow with the events
    seen) -- results must not change and the log must stay within a small multiple of the live windows."""
    from oracle.tgn_ref import TGNMemoryRef
    from tgm_amd.nn import IdentityMessage, LastAggregator, MeanAggregator, TGNMemory
    from tgm_amd.synth import make_stream

    st = make_stream('review', seed=5, num_edges=5200, n_src=900, n_dst=120)  # (the CPU oracle is this test's time: 5 200 events, the last 1 100 in eval mode)
    # strictly increasing times: a node's events inside a batch never tie (the reference leaves ties unspecified)
    ts = st.ts[0] + torch.arange(st.num_edges) * 300
    N, D, M, T_ = st.num_nodes, 16, 100, 100
    torch.manual_seed(3)
    mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator() if aggr == 'last' else MeanAggregator()).to(DEV).train()
    params = {k: v.detach().cpu().clone() for k, v in mem.state_dict().items() if k not in ('memory', 'last_update', '_assoc')}
    ref = TGNMemoryRef(N, D, M, T_, params, aggr)
    if log_cap:
        mem._log_cap_min = log_cap
    g = torch.Generator().manual_seed(9)
    for b, lo in enumerate(range(0, st.num_edges, bs)):
        hi = min(lo + bs, st.num_edges)
        if lo >= 4096 and mem.training:
            mem.eval()
            ref.eval()
            close(mem.memory.cpu(), ref.memory, 'flush')
        src, dst, t, raw = st.src[lo:hi], st.dst[lo:hi], ts[lo:hi], st.edge_x[lo:hi]
        neg = torch.randint(900, N, (hi - lo,), generator=g, dtype=torch.int32)
        n_id = torch.unique(torch.cat([src, dst, neg]))
        with _grad_mode(inference):
            z, lu = mem(n_id.to(DEV))
        z_ref, lu_ref = ref.forward(n_id.long())
        close(z.cpu(), z_ref, f'b{b} z')
        assert torch.equal(lu.cpu(), lu_ref)
        with _grad_mode(inference):
            mem.update_state(src.to(DEV), dst.to(DEV), t.to(DEV), raw.to(DEV))
        ref.update_state(src, dst, t, raw)
        if log_cap and mem.training:
            live = int(mem._st_cnt[0].sum()) + int(mem._st_cnt[1].sum())
            assert mem._log_other.numel() <= max(log_cap, 4 * (live + 2 * bs)), (mem._log_other.numel(), live)
    close(mem.memory.cpu(), ref.memory, 'final memory')
    assert torch.equal(mem.last_update.cpu(), ref.last_update)


@pytest.mark.parametrize('inference', [False, True])
def test_graph_attention_embedding_matches_restatement(inference):
    from oracle.tgn_ref import graph_attention_embedding_ref
    from tgm_amd.nn import GraphAttentionEmbedding, Time2Vec

    torch.manual_seed(0)
    U, E, M, D, T_, emb = 300, 4000, 100, 16, 100, 100
    enc = GraphAttentionEmbedding(M, emb, D, Time2Vec(T_)).to(DEV).eval()
    x = torch.randn(U, M)
    last_update = torch.randint(1_000_000, 2_000_000, (U,))
    edge_index = torch.stack([torch.randint(0, U, (E,)), torch.randint(0, 40, (E,))])  # few targets: long segments
    edge_index[1, :500] = torch.randint(0, U, (500,))
    t = torch.randint(0, 1_000_000, (E,))
    msg = torch.rand(E, D)
    with _grad_mode(inference):
        out = enc(x.to(DEV), last_update.to(DEV), edge_index.to(DEV), t.to(DEV), msg.to(DEV))
    ref = graph_attention_embedding_ref({k: v.cpu() for k, v in enc.state_dict().items()}, x, last_update, edge_index, t, msg)
    assert out.shape == (U, emb)
    close(out.cpu(), ref, 'graph attention embedding')


@pytest.mark.parametrize('fused', ['1', '0'])
@pytest.mark.parametrize('U,E,hubs', [(300, 4000, 40), (7000, 15000, 0), (50, 6000, 2), (1, 7, 0), (9000, 3, 0), (40, 23000, 1), (5000, 30000, 3), (2049, 70000, 0),
                                      (16000, 20000, 0), (16001, 20000, 0), (40000, 9000, 5)])
def test_counting_grouping_equals_the_segment_sort_path(U, E, hubs, fused, monkeypatch):
    """tgmx_tconv_forward groups a batch's edges by target with counts + an atomic cursor and lets the attention sort every segment
    by edge id (ascending edge id = the stable order of a sort by target): the embedding must equal the segment-sort path's BIT FOR
    BIT, also for hubs with more incoming edges than the attention's LDS buffer holds (> 1024: runs ranked in LDS, then merged pairwise --
    3 000 edges = 2 passes, 22 500 = 5 passes over a ragged last run),
    repeatedly (the count buffer is left zero), and after the batch shape changed.  Up to 16 000 targets the scan and the placement are ONE
    launch (every workgroup scans the counts itself; the attention clears them), beyond that -- and with TGMX_TCONV_GROUP_FUSED=0 -- two."""
    monkeypatch.setenv('TGMX_TCONV_GROUP_FUSED', fused)
    from tgm_amd.nn import GraphAttentionEmbedding, Time2Vec

    torch.manual_seed(U + E)
    M, D, T_, emb = 100, 16, 100, 100
    enc = GraphAttentionEmbedding(M, emb, D, Time2Vec(T_)).to(DEV).eval()

    def inputs(U, E, hubs, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(U, M, generator=g)
        lu = torch.randint(1_000_000, 2_000_000, (U,), generator=g)
        tgt = torch.randint(0, hubs, (E,), generator=g) if hubs else torch.randint(0, U, (E,), generator=g)
        if hubs and E > 500:
            tgt[:500] = torch.randint(0, U, (500,), generator=g)
        ei = torch.stack([torch.randint(0, U, (E,), generator=g), tgt])
        return [v.to(DEV) for v in (x, lu, ei, torch.randint(0, 1_000_000, (E,), generator=g), torch.rand(E, D, generator=g))]

    with torch.no_grad():
        for rep, (u, e, h) in enumerate([(U, E, hubs), (U, E, hubs), (max(U // 2, 1), max(E // 3, 1), hubs and 1), (U, E, hubs)]):
            args = inputs(u, e, h, 7 * rep + 1)
            monkeypatch.setenv('TGMX_TCONV_COUNTING', '1')
            z = enc(*args)
            monkeypatch.setenv('TGMX_TCONV_COUNTING', '0')
            z_ref = enc(*args)
            assert torch.equal(z, z_ref), f'round {rep}: max |d| = {(z - z_ref).abs().max().item():.3e}'
            cnt = enc.conv._tgt_count
            assert int(cnt.abs().sum().item()) == 0, 'the per-target counts must be left zero'


@pytest.mark.parametrize('one_launch', ['1', '0'])
@pytest.mark.parametrize('U,E', [(1, 1), (5, 0), (300, 40), (7000, 15000), (40_000, 3), (32768, 16384), (3, 16385)])
def test_segment_sort_matches_stable_argsort(U, E, one_launch, monkeypatch):
    """``tgmx_segment_sort`` (TransformerConv's incoming-edge grouping) == torch's stable argsort + searchsorted bounds."""
    from tgm_amd.nn.tgn import TransformerConv

    monkeypatch.setenv('TGMX_SEGSORT_SMALL', one_launch)
    conv = TransformerConv(8, 4, heads=1, dropout=0.0, edge_dim=2).to(DEV).eval()
    g = torch.Generator().manual_seed(U + E)
    tgt = torch.randint(0, U, (E,), generator=g).to(DEV)
    order, lo, hi = conv._incoming_segments(tgt, U)
    srt, ref_order = torch.sort(tgt, stable=True)
    ids = torch.arange(U, device=DEV)
    assert torch.equal(order, ref_order)
    assert torch.equal(lo, torch.searchsorted(srt, ids, right=False)) and torch.equal(hi, torch.searchsorted(srt, ids, right=True))


@pytest.mark.parametrize('U,seeds,k', [(9000, 1536, 10), (9000, 1536, 1), (300, 4096, 4), (300, 4097, 3), (32768, 1638, 10), (40, 1, 16384), (5, 2, 3), (9000, 1700, 10)])
@pytest.mark.parametrize('one_launch', ['1', '0'])
def test_segment_sort_of_seed_major_edge_lists(U, seeds, k, one_launch, monkeypatch):
    """Keys that come in runs (the k slots of a seed share its target; ragged: pad slots are compacted away) take the one-launch
    kernel's run path (<= 4096 runs), anything else its radix sort; the default is the rocPRIM chain: same result."""
    from tgm_amd.nn.tgn import TransformerConv

    monkeypatch.setenv('TGMX_SEGSORT_SMALL', one_launch)
    conv = TransformerConv(8, 4, heads=1, dropout=0.0, edge_dim=2).to(DEV).eval()
    g = torch.Generator().manual_seed(U + seeds + k)
    per_seed = torch.randint(0, U, (seeds,), generator=g)
    keep = torch.rand(seeds, k, generator=g) < 0.8
    keep[0, 0] = True
    tgt = per_seed[:, None].expand(seeds, k)[keep].to(DEV)
    order, lo, hi = conv._incoming_segments(tgt, U)
    srt, ref_order = torch.sort(tgt, stable=True)
    ids = torch.arange(U, device=DEV)
    assert torch.equal(order, ref_order)
    assert torch.equal(lo, torch.searchsorted(srt, ids, right=False)) and torch.equal(hi, torch.searchsorted(srt, ids, right=True))


def _tgn_stream_pipeline(pool, seed=8):
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook
    from tgm_amd.synth import make_stream

    st = make_stream('review', seed=seed, num_edges=6144, n_src=700, n_dst=100)
    ts = st.ts[0] + torch.arange(st.num_edges) * 300
    dg = DGraph(DGData.from_raw(ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', RandomNegativeEdgeSamplerHook(700, st.num_nodes, seed=4))
    hm.register('k', RecencyNeighborHook(st.num_nodes, [10], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], validate='deferred'))
    hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
    return st, hm, DGDataLoader(dg, batch_size=512, hook_manager=hm, output_pool=pool)


def _reference_glue(batch, k):
    """examples/linkproppred/tgn.py:80-92, verbatim in torch ops"""
    nbr = batch.nbr_nids[0].flatten()
    mask = nbr != -1
    seeds = torch.cat([batch.edge_src, batch.edge_dst, batch.neg]).repeat_interleave(k)
    edge_index = torch.stack([batch.global_to_local(seeds[mask]), batch.global_to_local(nbr[mask])]).long()
    return edge_index, batch.nbr_edge_time[0].flatten()[mask], batch.nbr_edge_x[0].flatten(0, -2)[mask]


def test_sampled_edge_list_equals_the_reference_loop_s_torch_glue():
    from tgm_amd.nn import sampled_edge_list

    st, hm, loader = _tgn_stream_pipeline(pool=1)
    with hm.activate('k'):
        for b, batch in enumerate(loader):
            ei, et, ex = sampled_edge_list(batch)
            ei_ref, et_ref, ex_ref = _reference_glue(batch, 10)
            assert ei.dtype == torch.int64 and torch.equal(ei, ei_ref), b
            assert torch.equal(et, et_ref) and torch.equal(ex, ex_ref), b
    assert b == 11


@pytest.mark.parametrize('aggr', ['last', 'mean'])
def test_reuse_forward_commits_the_same_state(aggr):
    """TGNMemory.reuse_forward: update_state copies the rows the forward produced (tgn.py:165-177 recomputes them from the
    same state) -- memory, last_update, the stores and every forward output stay bit-identical to the default path, with
    fresh and with pooled sampler outputs; a node outside the forward's n_id is reported by check()."""
    from tgm_amd.nn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, MeanAggregator, TGNMemory, sampled_edge_list

    runs = {}
    for reuse in (False, True):
        st, hm, loader = _tgn_stream_pipeline(pool=1 if reuse else 0)
        N, D, M, T_ = st.num_nodes, 16, 32, 20
        torch.manual_seed(5)
        mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator() if aggr == 'last' else MeanAggregator()).to(DEV).train()
        enc = GraphAttentionEmbedding(M, 32, D, mem.time_enc).to(DEV).eval()
        mem.reuse_forward = reuse
        outs = []
        with hm.activate('k'), torch.no_grad():
            for batch in loader:
                ei, et, ex = sampled_edge_list(batch) if reuse else _reference_glue(batch, 10)
                z, lu = mem(batch.unique_nids)
                outs.append((z.clone(), lu.clone(), enc(z, lu, ei, et, ex).clone()))
                mem.update_state(batch.edge_src, batch.edge_dst, batch.edge_time, batch.edge_x)
        mem.check()
        runs[reuse] = (outs, mem.memory.clone(), mem.last_update.clone(), mem)
    for (a, b) in zip(runs[False][0], runs[True][0]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert torch.equal(runs[False][1], runs[True][1]) and torch.equal(runs[False][2], runs[True][2])
    mem = runs[True][3]
    mem(torch.tensor([1, 2, 3], dtype=torch.int32, device=DEV))
    mem.update_state(torch.tensor([1, 699], dtype=torch.int32, device=DEV), torch.tensor([2, 3], dtype=torch.int32, device=DEV),
                     torch.tensor([10**9, 10**9 + 1], device=DEV), torch.rand(2, 16, device=DEV))
    with pytest.raises(RuntimeError, match='not in the preceding forward'):
        mem.check()


def test_prefetching_loader_and_edge_list_hook_yield_the_same_batches():
    """DGDataLoader(prefetch=1): batch i is handed out after batch i + 1 was enqueued; DeduplicationHook and
    SampledEdgeListHook learn their output sizes through deferred finalizers.  Every tensor equals the plain loader's
    (fresh tensors, no prefetch) and the reference loop's torch glue."""
    from tgm_amd import DGDataLoader
    from tgm_amd.hooks import SampledEdgeListHook

    st, hm_a, plain = _tgn_stream_pipeline(pool=0)
    _, hm_b, _ = _tgn_stream_pipeline(pool=0)
    hm_b.register('k', SampledEdgeListHook(hop=0))
    ahead = DGDataLoader(plain.dgraph, batch_size=512, hook_manager=hm_b, output_pool=2, prefetch=1)
    with pytest.raises(ValueError, match='output_pool'):
        DGDataLoader(plain.dgraph, batch_size=512, hook_manager=hm_b, output_pool=1, prefetch=1)
    n = 0
    with hm_a.activate('k'), hm_b.activate('k'):
        for ba, bb in zip(plain, ahead):
            assert ba._edge_lo == bb._edge_lo
            for name in ('edge_src', 'neg', 'unique_nids'):
                assert torch.equal(getattr(ba, name), getattr(bb, name)), (n, name)
            assert torch.equal(ba.nbr_nids[0], bb.nbr_nids[0]) and torch.equal(ba.nbr_edge_x[0], bb.nbr_edge_x[0])
            ei, et, ex = _reference_glue(ba, 10)
            assert torch.equal(bb.sampled_edge_index, ei) and torch.equal(bb.sampled_edge_time, et) and torch.equal(bb.sampled_edge_x, ex)
            assert torch.equal(bb.global_to_local(bb.edge_src), ba.global_to_local(ba.edge_src))
            n += 1
    assert n == 12


@pytest.mark.parametrize('n', [1025, 5000, 8192, 70_000])
def test_group_ids_large_matches_torch(n):
    """tgmx_group_ids_large (one stable radix sort + a finishing launch: what a 4096-edge batch's message store and commit use) against
    the torch formulation it replaces (tgm/nn/encoder/tgn.py:218-229: sort per role; :165-177: unique endpoints)."""
    from tgm_amd import _native

    lib = _native.load()
    g = torch.Generator().manual_seed(n)
    ids = torch.randint(0, max(n // 7, 3), (n,), generator=g, dtype=torch.int32).to(DEV)  # ~7 entries per id: every run length occurs
    ws = torch.empty(int(lib.tgmx_group_ids_workspace_bytes(n)), dtype=torch.uint8, device=DEV)
    srt = torch.empty(n, dtype=torch.int32, device=DEV)
    perm, lo, hi = (torch.empty(n, dtype=torch.int64, device=DEV) for _ in range(3))
    first = torch.empty(n, dtype=torch.uint8, device=DEV)
    _native.check(lib.tgmx_group_ids_large(ids.data_ptr(), n, srt.data_ptr(), perm.data_ptr(), lo.data_ptr(), hi.data_ptr(), first.data_ptr(),
                                           ws.data_ptr(), ws.numel(), _native.stream_ptr()), 'tgmx_group_ids_large')
    ref_s, ref_p = torch.sort(ids, stable=True)
    assert torch.equal(srt, ref_s) and torch.equal(perm, ref_p)
    assert torch.equal(lo, torch.searchsorted(ref_s, ref_s, right=False)) and torch.equal(hi, torch.searchsorted(ref_s, ref_s, right=True))
    ref_first = torch.ones(n, dtype=torch.uint8, device=DEV)
    ref_first[1:] = (ref_s[1:] != ref_s[:-1]).to(torch.uint8)
    assert torch.equal(first, ref_first)
    # every output is optional
    _native.check(lib.tgmx_group_ids_large(ids.data_ptr(), n, None, None, None, None, first.data_ptr(), ws.data_ptr(), ws.numel(), _native.stream_ptr()),
                  'tgmx_group_ids_large')
    assert torch.equal(first, ref_first)


@pytest.mark.parametrize('aggr,rider', [('last', '1'), ('mean', '1'), ('last', '0')])
def test_tgn_step_equals_the_three_module_calls(aggr, rider, monkeypatch):
    """TGNStep (tgmx_tgn_step: memory look-ahead -> embedding -> update_state as ONE native call) against memory(n_id), embedding(...),
    memory.update_state(...) on twin modules: z, last_update, z2 of every batch and the final memory / last_update tables equal BIT FOR BIT
    over 40 batches of a review-shaped stream, with a small message log so that compactions fall at different points of the sequence
    (before the forward here, between commit and store there), and the fallback (an evaluation-mode memory) gives the module path's results.
    rider: the commit as extra workgroups of the attention's launch (default) or as its own launch behind it (TGMX_TGN_COMMIT_RIDER=0)."""
    monkeypatch.setenv('TGMX_TGN_COMMIT_RIDER', rider)
    monkeypatch.setenv('TGMX_TGN_STORE_RIDER', rider)  # (likewise the batch's message store: two workgroups of the grouping's launch, or its own)
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook, SampledEdgeListHook
    from tgm_amd.nn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, MeanAggregator, TGNMemory, TGNStep
    from tgm_amd.synth import make_stream

    st = make_stream('review', seed=11, num_edges=40 * 256, n_src=1500, n_dst=200)
    N, D, M, T_, bs = st.num_nodes, 16, 100, 100, 256

    def build():
        torch.manual_seed(3)
        mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator() if aggr == 'last' else MeanAggregator()).to(DEV).train()
        mem.reuse_forward = True
        mem._log_cap_min = 1500  # (a batch stores 512 rows: a compaction every few batches)
        enc = GraphAttentionEmbedding(M, 100, D, mem.time_enc).to(DEV).eval()
        return mem, enc

    def loader():
        dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
        hm = HookManager(keys=['k'])
        hm.register('k', RandomNegativeEdgeSamplerHook(1500, N, seed=4))
        hm.register('k', RecencyNeighborHook(N, [10], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], validate='deferred',
                                             edge_features='by_id'))
        hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
        hm.register('k', SampledEdgeListHook(hop=0))
        return hm, DGDataLoader(dg, batch_size=bs, hook_manager=hm)

    (mem_a, enc_a), (mem_b, enc_b) = build(), build()
    step = TGNStep(mem_b, enc_b)
    hm, ld = loader()
    with hm.activate('k'), torch.no_grad():
        for b, batch in enumerate(ld):
            z, lu = mem_a(batch.unique_nids)
            z2 = enc_a(z, lu, batch.sampled_edge_index, batch.sampled_edge_time, batch.sampled_edge_x)
            mem_a.update_state(batch.edge_src, batch.edge_dst, batch.edge_time, batch.edge_x)
            y2, y, ylu = step.batch(batch)
            assert torch.equal(z, y) and torch.equal(lu, ylu) and torch.equal(z2, y2), f'batch {b}'
        mem_a.check()
        mem_b.check()
    # (the stream's first batch samples no neighbour yet: an empty edge list takes the module calls)
    assert step.fast_calls + step.fallback_calls == 40 and step.fast_calls >= 38
    assert torch.equal(mem_a.memory, mem_b.memory) and torch.equal(mem_a.last_update, mem_b.last_update)
    # the fallback: evaluation mode takes the three module calls
    mem_a.eval(), mem_b.eval()
    hm, ld = loader()
    with hm.activate('k'), torch.no_grad():
        batch = next(iter(ld))
        z, lu = mem_a(batch.unique_nids)
        z2 = enc_a(z, lu, batch.sampled_edge_index, batch.sampled_edge_time, batch.sampled_edge_x)
        mem_a.update_state(batch.edge_src, batch.edge_dst, batch.edge_time, batch.edge_x)
        y2, y, ylu = step.batch(batch)
    assert step.fast_calls + step.fallback_calls == 41 and torch.equal(z2, y2) and torch.equal(mem_a.memory, mem_b.memory)



def test_tgn_step_survives_module_calls_in_between():
    """The modules of a TGNStep used DIRECTLY between two steps (``memory(n_id)`` and the embedding on their own, ``reuse_forward`` toggled, a
    fallback batch): the step's cached static argument fields must not depend on what the modules do with their own argument blocks
    (they rewrite them in full on every module call).  Twin modules driven by the three module calls only give the reference."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook, SampledEdgeListHook
    from tgm_amd.nn import GraphAttentionEmbedding, IdentityMessage, LastAggregator, TGNMemory, TGNStep
    from tgm_amd.synth import make_stream

    st = make_stream('review', seed=12, num_edges=24 * 256, n_src=1500, n_dst=200)
    N, D, M, T_, bs = st.num_nodes, 16, 100, 100, 256

    def build():
        torch.manual_seed(3)
        mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator()).to(DEV).train()
        mem.reuse_forward = True
        return mem, GraphAttentionEmbedding(M, 100, D, mem.time_enc).to(DEV).eval()

    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', RandomNegativeEdgeSamplerHook(1500, N, seed=4))
    hm.register('k', RecencyNeighborHook(N, [10], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], validate='deferred', edge_features='by_id'))
    hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
    hm.register('k', SampledEdgeListHook(hop=0))
    (mem_a, enc_a), (mem_b, enc_b) = build(), build()
    step = TGNStep(mem_b, enc_b)
    with hm.activate('k'), torch.no_grad():
        for b, batch in enumerate(DGDataLoader(dg, batch_size=bs, hook_manager=hm)):
            z, lu = mem_a(batch.unique_nids)
            z2 = enc_a(z, lu, batch.sampled_edge_index, batch.sampled_edge_time, batch.sampled_edge_x)
            mem_a.update_state(batch.edge_src, batch.edge_dst, batch.edge_time, batch.edge_x)
            if b % 4 == 1:  # the modules on their own, on OTHER inputs than the step's: they rewrite their argument blocks (results discarded;
                some = batch.unique_nids[: max(1, batch.unique_nids.numel() // 3)]  # a pure look-ahead, nothing is committed)
                mem_b.reuse_forward = False
                zz, ll = mem_b(some)
                k_ = min(5, batch.sampled_edge_index.shape[1])
                if k_:
                    enc_b(torch.randn(int(batch.unique_nids.numel()), M, device=DEV), torch.zeros(int(batch.unique_nids.numel()), dtype=torch.long, device=DEV),
                          batch.sampled_edge_index[:, :k_], batch.sampled_edge_time[:k_], batch.sampled_edge_x[:k_])
                mem_b.reuse_forward = True
            y2, y, ylu = step.batch(batch)
            assert torch.equal(z, y) and torch.equal(lu, ylu) and torch.equal(z2, y2), f'batch {b}'
        mem_a.check()
        mem_b.check()
    assert step.fast_calls >= 20
    assert torch.equal(mem_a.memory, mem_b.memory) and torch.equal(mem_a.last_update, mem_b.last_update)


@pytest.mark.parametrize('inference', [False, True])
@pytest.mark.parametrize('emb,U,E,n_hot', [(100, 2000, 3000, 0), (128, 2000, 3000, 0), (128, 300, 4000, 40), (100, 50, 9000, 3), (6, 300, 2000, 20), (128, 40, 9000, 2)])
def test_graph_attention_embedding_walks_match_restatement(emb, U, E, n_hot, inference):
    """The three walks of ``tconv_attend_kernel`` against the restatement: two floats per lane (emb 100: C = 50 per head, cfg 3),
    four floats per lane (emb 128: C = 64), the generic walk (emb 6: C = 3) -- over segments that are all short (one wave, ids ranked
    in registers), a few hundred edges long (four waves, sorted in LDS) and thousands long (four waves, merge-sorted through memory)."""
    from oracle.tgn_ref import graph_attention_embedding_ref
    from tgm_amd.nn import GraphAttentionEmbedding, Time2Vec

    torch.manual_seed(emb * 7 + U)
    M, D, T_ = 100, 16, 100
    enc = GraphAttentionEmbedding(M, emb, D, Time2Vec(T_)).to(DEV).eval()
    x = torch.randn(U, M)
    last_update = torch.randint(1_000_000, 2_000_000, (U,))
    dst = torch.randint(0, n_hot, (E,)) if n_hot else torch.randint(0, U, (E,))
    if n_hot:
        dst[:500] = torch.randint(0, U, (500,))  # and a tail of short segments
    edge_index = torch.stack([torch.randint(0, U, (E,)), dst])
    t = torch.randint(0, 1_000_000, (E,))
    msg = torch.rand(E, D)
    with _grad_mode(inference):
        out = enc(x.to(DEV), last_update.to(DEV), edge_index.to(DEV), t.to(DEV), msg.to(DEV))
    ref = graph_attention_embedding_ref({k: v.cpu() for k, v in enc.state_dict().items()}, x, last_update, edge_index, t, msg)
    assert out.shape == (U, emb)
    close(out.detach().cpu(), ref, f'graph attention embedding (emb {emb}, U {U}, E {E})')

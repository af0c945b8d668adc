"""The lowered hook chain (DGDataLoader(output_pool=R) -> tgmx_pipeline_step: shard -> negatives generated in the seed
fetch -> recency sampler, pooled outputs) against (a) the same chain run hook by hook -- every produced tensor
bit-identical, negatives included -- and (b) the CPU oracle.  Reference behaviour restated:
tgm/data/loader.py:158-170 + tgm/hooks/hook_manager.py:139-168 + tgm/hooks/negatives/sampler.py:45-65 +
tgm/hooks/neighbors/recency.py:119-171."""
import logging

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _stream(E=5000, D=12, shape='wiki', **kw):
    from tgm_amd.synth import make_stream

    return make_stream(shape, seed=11, num_edges=E, edge_dim=D, **kw)


def _build(st, bs, num_nbrs, mode, pool, world=1, rank=0, neg=True, extra=None, directed=False):
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.dist import EdgeShardHook
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook

    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    hm = HookManager(keys=['k'])
    lo_dst = int(st.dst.min())
    if world > 1:
        hm.register('k', EdgeShardHook(rank, world))
        keys, tkeys = ['shard_src', 'shard_dst'], ['shard_time', 'shard_time']
        if neg:
            hm.register('k', RandomNegativeEdgeSamplerHook(lo_dst, st.num_nodes, seed=5, like='shard_dst', time_key='shard_time'))
    else:
        keys, tkeys = ['edge_src', 'edge_dst'], ['edge_time', 'edge_time']
        if neg:
            hm.register('k', RandomNegativeEdgeSamplerHook(lo_dst, st.num_nodes, seed=5))
    if neg:
        keys, tkeys = keys + ['neg'], tkeys + ['neg_time']
    hook = RecencyNeighborHook(st.num_nodes, num_nbrs, keys, tkeys, mode=mode, validate='deferred', directed=directed,
                               batch_size=bs if mode == 'csr' else None)  # fmt: skip
    hm.register('k', hook)
    for h in extra or []:
        hm.register('k', h)
    return hm, hook, DGDataLoader(dg, batch_size=bs, hook_manager=hm, output_pool=pool)


def _same(a, b, tag):
    if isinstance(a, torch.Tensor):
        assert isinstance(b, torch.Tensor) and a.dtype == b.dtype and a.shape == b.shape, f'{tag}: {a.dtype}{tuple(a.shape)} vs {b.dtype}{tuple(b.shape)}'
        assert torch.equal(a, b), tag
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), tag
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f'{tag}[{i}]')
    elif isinstance(a, dict):
        assert sorted(a) == sorted(b), tag
        for k in a:
            _same(a[k], b[k], f'{tag}[{k}]')
    else:
        assert a == b, tag


ATTRS = ['edge_src', 'edge_dst', 'edge_time', 'edge_x', 'seed_nids', 'seed_times', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x', 'seed_node_nbr_mask']


@pytest.mark.parametrize('mode', ['ring', 'csr'])
@pytest.mark.parametrize('world,rank', [(1, 0), (2, 0), (2, 1), (3, 1)])
@pytest.mark.parametrize('neg', [True, False])
def test_lowered_chain_equals_hook_by_hook(mode, world, rank, neg):
    st = _stream()
    bs, k = 96, [4, 3]  # 5000 % 96 = 8: a ragged last batch
    hm_a, hook_a, plain = _build(st, bs, k, mode, 0, world, rank, neg)
    hm_b, hook_b, pooled = _build(st, bs, k, mode, 3, world, rank, neg)
    attrs = ATTRS + (['neg', 'neg_time'] if neg else []) + (['shard_src', 'shard_dst', 'shard_time', 'shard_lo'] if world > 1 else [])
    with hm_a.activate('k'), hm_b.activate('k'):
        for epoch in range(2):
            n = 0
            for ba, bb in zip(plain, pooled):
                for name in attrs:
                    _same(getattr(ba, name), getattr(bb, name), f'epoch {epoch} batch {n} {name}')
                assert ba._edge_lo == bb._edge_lo
                n += 1
            assert n == len(plain) == len(pooled)
            hm_a.reset_state()
            hm_b.reset_state()
        assert pooled._compiled[1] is not None and pooled._compiled[1].n_lowered == (1 if world > 1 else 0) + (1 if neg else 0) + 1
    hook_a.check()
    hook_b.check()


@pytest.mark.parametrize('mode', ['ring', 'csr'])
@pytest.mark.parametrize('k,D,shape', [([70], 8, 'wiki'), ([10, 10], 16, 'review'), ([20, 20], 20, 'wiki'), ([3, 2, 2], 5, 'comment')])
def test_delta_feature_writes_equal_full_writes(mode, k, D, shape):
    """Pooled outputs are persistent, so the lookups rewrite a feature row only from its leftmost slot that changes
    (tgmx_recency_step_t.out_valid).  Every kernel that emits rows -- one wave per seed with k <= 64 and k > 64, the packed
    narrow-row groups, the fused hop 0 + 1 launch, three hops -- against fresh tensors per batch (every slot written), over two
    epochs with a reset in between (rows that held neighbors become all-pad again) and a ragged last batch."""
    st = _stream(E=3000, D=D, shape=shape)
    bs = 128
    hm_a, hook_a, plain = _build(st, bs, k, mode, 0)
    hm_b, hook_b, pooled = _build(st, bs, k, mode, 1)
    with hm_a.activate('k'), hm_b.activate('k'):
        for epoch in range(2):
            for n, (ba, bb) in enumerate(zip(plain, pooled)):
                for name in ('nbr_nids', 'nbr_edge_time', 'nbr_edge_x'):
                    _same(getattr(ba, name), getattr(bb, name), f'epoch {epoch} batch {n} {name}')
            hm_a.reset_state()
            hm_b.reset_state()
    cp = pooled._compiled[1]
    assert cp is not None and cp._delta and cp._sets and all(sl.valid for os_ in cp._sets for sl in os_.views.values()), 'the pooled loader did not take the delta path'
    hook_a.check()
    hook_b.check()


def test_delta_feature_writes_fuzz():
    """Random shapes through the pooled loader (delta feature writes) vs fresh tensors per batch: hop counts, k (incl. > 64), feature
    widths that are and are not multiples of 4, batch sizes that put the ring update on each of its plans, both modes, pool sizes
    1-3.  TGMX_FUZZ=<n> runs n configurations (default 12)."""
    import os
    import random

    rng = random.Random(2024)
    for it in range(int(os.environ.get('TGMX_FUZZ', '12'))):
        mode = rng.choice(['ring', 'ring', 'csr'])
        hops = rng.choice([1, 2, 2, 3])
        k = [rng.choice([1, 2, 3, 5, 8, 10, 16, 20, 33, 70]) for _ in range(hops)]
        while sum(1 for _ in k) > 1 and __import__('math').prod(k) > 4000:
            k[k.index(max(k))] = 4
        D = rng.choice([0, 1, 3, 4, 8, 12, 16, 20, 43])
        bs = rng.choice([7, 64, 200, 513, 700])
        shape = rng.choice(['wiki', 'review', 'comment'])
        pool = rng.choice([1, 1, 2, 3])
        st = _stream(E=rng.choice([900, 2500]), D=D, shape=shape)
        hm_a, hook_a, plain = _build(st, bs, k, mode, 0)
        hm_b, hook_b, pooled = _build(st, bs, k, mode, pool)
        tag = f'config {it}: mode={mode} k={k} D={D} bs={bs} {shape} pool={pool}'
        with hm_a.activate('k'), hm_b.activate('k'):
            for epoch in range(2):
                for n, (ba, bb) in enumerate(zip(plain, pooled)):
                    for name in ('nbr_nids', 'nbr_edge_time', 'nbr_edge_x'):
                        _same(getattr(ba, name), getattr(bb, name), f'{tag} epoch {epoch} batch {n} {name}')
                hm_a.reset_state()
                hm_b.reset_state()
        hook_a.check()
        hook_b.check()


def test_lookup_accounting_matches_torch():
    """tgmx_lookup_accounting (the byte model's counts of a timed launch as partial sums) against torch reductions."""
    from tgm_amd import _native

    lib = _native.load()
    g = torch.Generator().manual_seed(4)
    for rows, k in ((1, 1), (600, 20), (12_000, 20), (70_001, 3)):
        ids = torch.randint(-1, 50, (rows, k), generator=g, dtype=torch.int32).to(DEV)
        sp = torch.randint(0, k + 1, (2, rows), generator=g, dtype=torch.int32).to(DEV)
        parts = torch.full((_native.ACCOUNTING_PARTIALS, 3), -7, dtype=torch.int64, device=DEV)
        _native.check(lib.tgmx_lookup_accounting(ids.data_ptr(), ids.numel(), sp[1].data_ptr(), sp[0].data_ptr(), rows, parts.data_ptr(),
                                                 _native.stream_ptr()), 'tgmx_lookup_accounting')
        got = parts.sum(0).tolist()
        assert got == [int((ids != -1).sum()), int(torch.maximum(sp[0], sp[1]).sum()), int(sp[0].sum())], (rows, k, got)
        _native.check(lib.tgmx_lookup_accounting(ids.data_ptr(), ids.numel(), None, None, rows, parts.data_ptr(), _native.stream_ptr()), 'accounting')
        assert parts.sum(0).tolist() == [int((ids != -1).sum()), 0, 0]


def test_lowered_chain_vs_oracle_and_pool_recycling():
    """Pooled outputs against the CPU restatement of the reference (oracle/ring_port.py), with the comment-like
    non-bipartite shape (timestamp ties inside batches, self loops possible); the tensors of batch i are recycled by
    batch i + R and not before."""
    from oracle.ring_port import RingSamplerCPU

    st = _stream(E=6000, D=16, shape='comment', n_src=300, t_hi=1_100_000_500)
    bs, k, R = 256, [5, 5], 2
    hm, hook, loader = _build(st, bs, k, 'ring', R)
    ref = RingSamplerCPU(st.num_nodes, k, 16)
    held = []
    with hm.activate('k'):
        for b, batch in enumerate(loader):
            lo, hi = b * bs, min((b + 1) * bs, st.num_edges)
            neg = batch.neg.cpu()
            hops = ref.step(torch.cat([st.src[lo:hi], st.dst[lo:hi], neg]), torch.cat([st.ts[lo:hi]] * 3), st.src[lo:hi], st.dst[lo:hi],
                            st.ts[lo:hi], st.edge_x[lo:hi])  # fmt: skip
            for h, (s_n, s_t, o_i, o_t, o_x) in enumerate(hops):
                assert torch.equal(batch.seed_nids[h].cpu(), s_n) and torch.equal(batch.seed_times[h].cpu(), s_t)
                assert torch.equal(batch.nbr_nids[h].cpu(), o_i), f'batch {b} hop {h}'
                assert torch.equal(batch.nbr_edge_time[h].cpu(), o_t) and torch.equal(batch.nbr_edge_x[h].cpu(), o_x)
            held.append((batch.nbr_nids[1], batch.nbr_nids[1].clone()))
            if hi - lo == bs and b >= 1 and held[b - 1][0].shape == held[b][0].shape:
                assert torch.equal(*held[b - 1]), 'a pooled tensor changed before R more batches were produced'
                if b >= R and held[b - R][0].shape == held[b][0].shape:
                    assert held[b - R][0].data_ptr() == held[b][0].data_ptr()  # recycled after exactly R batches
    hook.check()


def test_hooks_behind_the_lowered_prefix_still_run():
    """A hook the native step does not cover (here: a DeduplicationHook with an id suffix) runs as usual behind the lowered prefix."""
    from tgm_amd.hooks import DeduplicationHook

    st = _stream(E=3000, D=8)
    hm_a, _, plain = _build(st, 128, [5], 'ring', 0, extra=[DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids'], id='t')])
    hm_b, _, pooled = _build(st, 128, [5], 'ring', 2, extra=[DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids'], id='t')])
    with hm_a.activate('k'), hm_b.activate('k'):
        for ba, bb in zip(plain, pooled):
            _same(ba.unique_nids_t, bb.unique_nids_t, 'unique_nids')
            _same(ba.global_to_local_t(ba.edge_src), bb.global_to_local_t(bb.edge_src), 'global_to_local')
        assert pooled._compiled[1].n_lowered == 2


def test_sync_validation_is_lowered_too_and_raises_per_batch():
    """validate='sync' (the hook's default, the reference's raise-per-call behaviour) through the lowered chain: same tensors
    as hook by hook, one status read per batch."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RecencyNeighborHook

    st = _stream(E=600, D=4)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    outs = []
    for pool in (0, 2):
        hm = HookManager(keys=['k'])
        hm.register('k', RecencyNeighborHook(st.num_nodes, [3], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time']))  # validate='sync'
        loader = DGDataLoader(dg, batch_size=100, hook_manager=hm, output_pool=pool)
        with hm.activate('k'):
            outs.append([(b.nbr_nids[0].clone(), b.nbr_edge_x[0].clone()) for b in loader])
            assert (loader._compiled is not None and loader._compiled[1] is not None) == (pool > 0)
    _same(outs[0], outs[1], 'sync mode, lowered vs hook by hook')
    # a hook built for too few nodes: the stream's ids are out of range -> ValueError at the first batch, in both paths
    for pool in (0, 2):
        hm2 = HookManager(keys=['k'])
        hm2.register('k', RecencyNeighborHook(10, [3], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time']))
        with hm2.activate('k'), pytest.raises(ValueError):
            next(iter(DGDataLoader(dg, batch_size=100, hook_manager=hm2, output_pool=pool)))


@pytest.mark.parametrize('bs', [100, 800, 2600])  # update plans: rider + commit (m = 200), chunked placement (m = 1600), radix sort (m = 5200)
@pytest.mark.parametrize('validate', ['sync', 'deferred'])
def test_bad_seeds_leave_the_rings_untouched(bs, validate):
    """The reference validates the seeds before it changes anything (recency.py:173-237 runs before _update).  Here lookups and
    update are ONE call.  validate='sync' (raise per call): when the lookups flag a seed, every kernel that writes ring state
    returns untouched (guard_seed_errors), so the caller that raises sees unchanged state.  validate='deferred': the error is only
    REPORTED later, so nothing is guarded -- the status word is sticky until check(), and a guard on it would silently drop the
    update of every later batch; the batch's edges (which are valid: only a seed was bad) are appended as usual."""
    from tgm_amd import DGData, DGraph
    from tgm_amd.hooks import RecencyNeighborHook

    st = _stream(E=4 * bs + 50, D=8)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    make = lambda: RecencyNeighborHook(st.num_nodes, [4, 2], ['edge_src', 'extra'], ['edge_time', 'extra_t'], validate=validate)
    hook, twin = make(), make()  # the twin sees the same batches with good seeds only
    good = lambda b: (setattr(b, 'extra', b.edge_dst.clone()), setattr(b, 'extra_t', b.edge_time.clone()))
    for i in range(2):
        for h in (hook, twin):
            b = dg.slice_events(i * bs, (i + 1) * bs).materialize()
            good(b)
            h(dg, b)
    hook.check()
    state = lambda h: (h._ring.clone(), h._write_pos.clone(), h._ring_x.clone())
    def same(x, y):  # feature rows of slots that hold no record are uninitialised memory: compare the rows of real records
        live = (x[0][:, 0] & 0xFFFFFFFF) < 0x80000000
        return torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and torch.equal(x[2][live], y[2][live])

    before = state(hook)
    b = dg.slice_events(2 * bs, 3 * bs).materialize()
    good(b)
    b.extra[bs // 2] = st.num_nodes + 5  # one seed out of range
    with pytest.raises(ValueError):
        hook(dg, b)
        hook.check()
    if validate == 'sync':
        assert same(state(hook), before), 'a call that raised changed the rings'
        good(b)  # the same batch with valid seeds goes through and does change the state
        hook(dg, b)
        hook.check()
    good(b)
    twin(dg, b)
    assert same(state(hook), state(twin))
    for h in (hook, twin):  # and the batches after it are appended in both modes (nothing stays blocked)
        b = dg.slice_events(3 * bs, 4 * bs).materialize()
        good(b)
        h(dg, b)
    hook.check()
    assert same(state(hook), state(twin)) and not same(state(hook), before)
    assert not torch.equal(hook._write_pos, before[1])


def test_csr_mode_refuses_batches_off_the_indexed_schedule():
    """a batch that does not start on a boundary of the schedule the index was built for must not silently
    return wrong neighbours."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RecencyNeighborHook

    st = _stream(E=2000, D=4)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    for pool in (0, 2):
        hook = RecencyNeighborHook(st.num_nodes, [3], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time'], mode='csr', batch_size=128,
                                   validate='deferred')  # fmt: skip
        hm = HookManager(keys=['k'])
        hm.register('k', hook)
        with hm.activate('k'):
            it = iter(DGDataLoader(dg, batch_size=128, hook_manager=hm, output_pool=pool))
            next(it), next(it)
            with pytest.raises(ValueError, match='not a boundary'):
                next(iter(DGDataLoader(dg.slice_events(300, 2000), batch_size=128, hook_manager=hm, output_pool=pool)))
            # an aligned split is fine
            next(iter(DGDataLoader(dg.slice_events(384, 2000), batch_size=128, hook_manager=hm, output_pool=pool)))


def test_out_of_order_batches_warn_once(caplog):
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RecencyNeighborHook

    st = _stream(E=1000, D=4)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', RecencyNeighborHook(st.num_nodes, [3], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time']))
    loader = DGDataLoader(dg, batch_size=100, hook_manager=hm)
    with hm.activate('k'), caplog.at_level(logging.WARNING, logger='tgm_amd.hooks.recency'):
        loader(500)
        loader(600)
        assert not caplog.records
        loader(100)  # behind the previous batch, state not reset
        loader(0)
        assert len([r for r in caplog.records if 'chronological' in r.getMessage()]) == 1
        hm.reset_state()
        loader(0)
        assert len(caplog.records) == 1


def test_masks_are_fresh_tensors_per_batch():
    """recency.py:221-224 hands out new index tensors per batch: an in-place edit must not leak into the next batch."""
    st = _stream(E=600, D=4)
    hm, _, loader = _build(st, 100, [3], 'ring', 0)
    with hm.activate('k'):
        it = iter(loader)
        b0 = next(it)
        b0.seed_node_nbr_mask['edge_dst'].fill_(-7)
        b1 = next(it)
        assert torch.equal(b1.seed_node_nbr_mask['edge_dst'].cpu(), torch.arange(100, 200))
        assert torch.equal(b1.seed_node_nbr_mask['neg'].cpu(), torch.arange(200, 300))


def test_negatives_not_used_as_seeds_are_not_lowered():
    """A negative hook whose ids the sampler does not seed from cannot ride in the lowered step (the ids are drawn in the seed
    fetch): the chain then runs hook by hook, and `neg` is still produced."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook

    st = _stream(E=800, D=4)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', RandomNegativeEdgeSamplerHook(int(st.dst.min()), st.num_nodes, seed=1))
    hm.register('k', RecencyNeighborHook(st.num_nodes, [3], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time'], validate='deferred'))
    loader = DGDataLoader(dg, batch_size=100, hook_manager=hm, output_pool=2)
    with hm.activate('k'):
        b = next(iter(loader))
        assert loader._compiled[1] is None
        assert b.neg.shape == (100,) and int(b.neg.min()) >= int(st.dst.min()) and b.seed_nids[0].shape == (200,)


def test_lowered_tgn_tail_dedup_and_edge_list():
    """The chain negatives -> sampler -> DeduplicationHook -> SampledEdgeListHook as ONE tgmx_pipeline_step (post block): every
    tensor equals the hook-by-hook chain's, with and without the loader running a batch ahead, for hop 0 of a two-hop sampler
    and with the neighbor ids of both hops deduplicated."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook, SampledEdgeListHook

    st = _stream(E=4000, D=8, shape='review', n_src=400, n_dst=80)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)

    def chain():
        hm = HookManager(keys=['k'])
        hm.register('k', RandomNegativeEdgeSamplerHook(400, st.num_nodes, seed=9))
        hm.register('k', RecencyNeighborHook(st.num_nodes, [6, 3], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], validate='deferred'))
        hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
        hm.register('k', SampledEdgeListHook(hop=0))
        return hm

    hm_a, hm_b, hm_c = chain(), chain(), chain()
    plain = DGDataLoader(dg, batch_size=256, hook_manager=hm_a)
    pooled = DGDataLoader(dg, batch_size=256, hook_manager=hm_b, output_pool=1)
    ahead = DGDataLoader(dg, batch_size=256, hook_manager=hm_c, output_pool=3, prefetch=2)
    names = ('neg', 'unique_nids', 'sampled_edge_index', 'sampled_edge_time', 'sampled_edge_x')
    with hm_a.activate('k'), hm_b.activate('k'), hm_c.activate('k'):
        for n, (a, b, c) in enumerate(zip(plain, pooled, ahead)):
            for other in (b, c):
                for name in names:
                    _same(getattr(a, name), getattr(other, name), f'batch {n} {name}')
                _same(a.nbr_nids, other.nbr_nids, f'batch {n} nbr_nids')
                _same(a.global_to_local(a.edge_dst), other.global_to_local(other.edge_dst), 'global_to_local')
        assert n == 15
        assert pooled._compiled[1].n_lowered == 4 and ahead._compiled[1].n_lowered == 4


def test_edge_features_by_id_with_the_tgn_tail():
    """RecencyNeighborHook(edge_features='by_id') in front of DeduplicationHook -> SampledEdgeListHook under the DEFAULT loader
    (that combination once failed in the lowered post block, which read the dense [S, k, D] copies the mode does not make).
    Round 4: the whole chain lowers (4 hooks) and the post block writes the list's feature rows from the resident store by edge id
    (tgmx_tgn_edge_list_by_id); hook by hook the edge-list hook takes the same entry point.  Every tensor equals the dense chain's."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook, SampledEdgeListHook

    st = _stream(E=3000, D=8, shape='review', n_src=300, n_dst=60)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)

    def chain(features):
        hm = HookManager(keys=['k'])
        hm.register('k', RandomNegativeEdgeSamplerHook(300, st.num_nodes, seed=9))
        hm.register('k', RecencyNeighborHook(st.num_nodes, [5, 3], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'],
                                             edge_features=features))
        hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
        hm.register('k', SampledEdgeListHook(hop=0))
        return hm

    hm_a, hm_b, hm_c = chain('dense'), chain('by_id'), chain('by_id')
    dense = DGDataLoader(dg, batch_size=200, hook_manager=hm_a)
    by_id = DGDataLoader(dg, batch_size=200, hook_manager=hm_b)  # the default pool: lowers
    by_id_hooks = DGDataLoader(dg, batch_size=200, hook_manager=hm_c, output_pool=0)
    names = ('neg', 'unique_nids', 'sampled_edge_index', 'sampled_edge_time', 'sampled_edge_x')
    with hm_a.activate('k'), hm_b.activate('k'), hm_c.activate('k'):
        for n, (a, b, c) in enumerate(zip(dense, by_id, by_id_hooks)):
            for other in (b, c):
                for name in names:
                    _same(getattr(a, name), getattr(other, name), f'batch {n} {name}')
                _same(a.nbr_nids, other.nbr_nids, f'batch {n} nbr_nids')
                for h in range(2):
                    _same(a.nbr_edge_x[h], other.nbr_edge_x[h], f'batch {n} nbr_edge_x[{h}]')
        assert n == 14
        assert dense._compiled[1].n_lowered == 4 and by_id._compiled[1].n_lowered == 4


# ---- the loader's DEFAULT: lowered chain, fresh-tensor semantics from liveness-checked output sets (round 3) --------------------
def _default_loader(st, bs, k, mode='ring', validate='sync', D=None, num_nodes=None):
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook

    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    hm = HookManager(keys=['k'])
    N = num_nodes or st.num_nodes
    hm.register('k', RandomNegativeEdgeSamplerHook(int(st.dst.min()), N, seed=5))
    hook = RecencyNeighborHook(N, k, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], mode=mode, validate=validate,
                               batch_size=bs if mode == 'csr' else None)  # fmt: skip
    hm.register('k', hook)
    return dg, hm, hook, DGDataLoader(dg, batch_size=bs, hook_manager=hm)  # no output_pool=: the default


@pytest.mark.parametrize('mode', ['ring', 'csr'])
def test_default_loader_lowers_and_keeps_fresh_tensor_semantics(mode):
    """An unmodified script -- DGDataLoader(dg, batch_size, hook_manager=hm), RecencyNeighborHook's default validate='sync' -- runs
    the lowered chain.  Its tensors behave like the reference's fresh ones (recency.py:119-171 returns new gathers per call): a
    batch somebody still holds is never overwritten, whatever they hold of it."""
    st = _stream(E=3000, D=8)
    bs, k = 100, [5, 4]
    hm_a, hook_a, plain = _build(st, bs, k, mode, 0)
    plain_batches = []
    with hm_a.activate('k'):
        for b in plain:
            plain_batches.append(b)
    names = ('seed_nids', 'seed_times', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x', 'seed_node_nbr_mask', 'neg', 'neg_time')
    # (every scenario below starts from fresh hooks: the negatives' call counter and the rings then match the baseline's)
    dg, hm, hook, loader = _default_loader(st, bs, k, mode)
    with hm.activate('k'):
        kept = list(loader)  # every batch alive at once: nothing may be recycled
        cp = loader._compiled[1]
        assert cp is not None and cp._safe and cp.n_lowered == 2, 'the default loader did not lower the chain'
        for n, (a, b) in enumerate(zip(plain_batches, kept)):
            for name in names:
                _same(getattr(a, name), getattr(b, name), f'all alive, batch {n} {name}')
        del kept, b
    hook.check()
    dg, hm, hook, loader = _default_loader(st, bs, k, mode)
    with hm.activate('k'):
        # a loop that drops each batch before the next is produced stays on ONE set
        ptrs = set()
        for n, s0 in enumerate(loader._starts):
            b = loader(s0)
            ptrs.add(b.nbr_edge_x[1].data_ptr())
            for name in names:
                _same(getattr(plain_batches[n], name), getattr(b, name), f'dropped, batch {n} {name}')
            del b
        assert len(ptrs) == 1, f'{len(ptrs)} output sets used by a loop that holds no batch'
    hook.check()
    dg, hm, hook, loader = _default_loader(st, bs, k, mode)
    with hm.activate('k'):
        # `for batch in loader` holds batch i while batch i + 1 is produced: two sets alternate, contents stay right
        prev = None
        for n, b in enumerate(loader):
            if prev is not None:
                for name in names:
                    _same(getattr(plain_batches[n - 1], name), getattr(prev, name), f'for-loop, previous batch {n - 1} {name}')
            prev = b
        assert len(loader._compiled[1]._sets) == 2
        del prev, b
    hook.check()
    dg, hm, hook, loader = _default_loader(st, bs, k, mode)
    with hm.activate('k'):
        # what a consumer may hold of a batch: a tensor, a view of one, a detached alias, a tensor autograd saved
        holders = []
        w = torch.ones(8, 1, device=DEV, requires_grad=True)
        for n, s0 in enumerate(loader._starts[:8]):
            b = loader(s0)
            ref = plain_batches[n]
            how = n % 4
            if how == 0:
                holders.append((b.nbr_nids[1], ref.nbr_nids[1]))
            elif how == 1:
                holders.append((b.nbr_edge_x[1][3:9], ref.nbr_edge_x[1][3:9]))
            elif how == 2:
                holders.append((b.nbr_edge_time[0].detach(), ref.nbr_edge_time[0]))
            else:
                y = (b.nbr_edge_x[0] @ w).sum()  # saves nbr_edge_x[0] for backward
                holders.append((y, ref.nbr_edge_x[0]))
            del b
        for n, (got, want) in enumerate(holders):
            if got.dim() == 0:
                (g,) = torch.autograd.grad(got, w)
                assert torch.allclose(g, want.sum((0, 1)).view(8, 1), rtol=1e-4), f'holder {n}: the tensor autograd saved was overwritten'
            else:
                _same(got, want, f'holder {n}')
    hook.check()


def test_default_loader_survives_in_place_modification():
    """Fresh tensors may be modified in place by their owner.  The persistent sets rely on pad slots staying zero (delta feature
    writes): an in-place torch op moves the buffer's version counter and the set is re-initialised before its next use."""
    st = _stream(E=2000, D=8)
    bs, k = 100, [6, 3]
    hm_a, _, plain = _build(st, bs, k, 'ring', 0)
    dg, hm, hook, loader = _default_loader(st, bs, k)
    with hm_a.activate('k'), hm.activate('k'):
        for n, (a, s0) in enumerate(zip(plain, loader._starts)):
            b = loader(s0)
            for name in ('nbr_nids', 'nbr_edge_time', 'nbr_edge_x'):
                _same(getattr(a, name), getattr(b, name), f'batch {n} {name}')
            if n % 3 == 1:
                b.nbr_edge_x[1].add_(1.0)  # pads are no longer zero
                b.nbr_nids[0].fill_(7)
                b.seed_node_nbr_mask['neg'].zero_()
            del b
    hook.check()


def test_default_sync_validation_raises_per_call_without_touching_state():
    """validate='sync' (the hook's default) through the lowered chain: the reference raises in the call that carries a bad seed,
    before it touches its state (recency.py:214-229).  The seeds are rows of the resident store, so the loader knows the
    offending edges up front: the batch that contains one raises ValueError, the others run, no device read per batch."""
    st = _stream(E=1500, D=4)
    N = st.num_nodes
    bad_edge = 777
    src = st.src.clone()
    src[bad_edge] = N + 3  # out of the hook's node range (the store itself allows it: DGData infers num_nodes from the ids)
    import dataclasses

    st2 = dataclasses.replace(st, src=src)
    dg, hm, hook, loader = _default_loader(st2, 100, [4, 2], num_nodes=N)
    with hm.activate('k'):
        for n, s0 in enumerate(loader._starts):
            if s0 <= bad_edge < s0 + 100:
                ring_before = hook._ring.clone()
                with pytest.raises(ValueError, match='Seed nodes must satisfy'):
                    loader(s0)
                assert torch.equal(ring_before, hook._ring), 'a refused batch changed the rings'
            else:
                loader(s0)
        cp = loader._compiled[1]
        assert cp._static_ok[0] and list(cp._static_ok[1]) == [bad_edge]
    hook.check()  # the device never saw the bad seed


def test_prefetch_settles_finalizers_for_hooks_that_need_them():
    """A hook ordered behind DeduplicationHook that READS unique_nids must find it even when the loader runs a batch ahead and
    the dedup result's size is still in flight (the finalizer is then run before that hook instead of one batch later)."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook
    from tgm_amd.hooks.base import StatelessHook

    class NeedsUnique(StatelessHook):
        _cls_requires = {'unique_nids'}
        _cls_produces = {'n_unique'}

        def __init__(self):
            super().__init__()
            self.__post_init__()

        def __call__(self, dg, batch):
            batch.n_unique = int(batch.unique_nids.numel())  # AttributeError before the fix
            return batch

    st = _stream(E=1200, D=4, shape='review', n_src=300, n_dst=60)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    for pool in (None, 0, 3):
        hm = HookManager(keys=['k'])
        hm.register('k', RandomNegativeEdgeSamplerHook(300, st.num_nodes, seed=9))
        hm.register('k', RecencyNeighborHook(st.num_nodes, [5], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], validate='deferred'))
        hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
        hm.register('k', NeedsUnique())
        loader = DGDataLoader(dg, batch_size=128, hook_manager=hm, output_pool=pool, prefetch=1)
        with hm.activate('k'):
            for b in loader:
                assert b.n_unique == b.unique_nids.numel() > 0

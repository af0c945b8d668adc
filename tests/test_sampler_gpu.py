"""GPU parity: the HIP sampler (through the C ABI, via the drop-in hook) against
the golden vectors recorded from the reference and against the CPU oracle.

Bit-exact bar: neighbor ids, timestamps and the copied feature rows.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _mk():
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RecencyNeighborHook
    from tgm_amd.hooks.base import StatelessHook

    class ReplayNegatives(StatelessHook):
        _cls_requires = {'edge_src', 'edge_dst', 'edge_time'}
        _cls_produces = {'neg', 'neg_time'}

        def __init__(self, neg):
            super().__init__()
            self.neg = neg
            self.base = 0
            self.__post_init__()

        def __call__(self, dg, batch):
            lo = self.base + batch._edge_lo
            batch.neg = self.neg[lo : lo + batch.edge_src.numel()].clone()
            batch.neg_time = batch.edge_time.clone()
            return batch

    return DGData, DGDataLoader, DGraph, HookManager, RecencyNeighborHook, ReplayNegatives


def _graph(a, lo, hi, edge_x):
    DGData, _, DGraph, *_ = _mk()
    T = torch.from_numpy
    d = DGData.from_raw(
        T(a['ts'][lo:hi]),
        torch.stack([T(a['src'][lo:hi]), T(a['dst'][lo:hi])], 1),
        None if edge_x is None else edge_x[lo:hi],
    )
    return DGraph(d, device=DEV)


def _compare(meta, a, b, batch, L, tag):
    for h in range(L):
        got = dict(
            seed_nids=batch.seed_nids[h], seed_times=batch.seed_times[h], nbr_nids=batch.nbr_nids[h],
            nbr_edge_time=batch.nbr_edge_time[h], nbr_edge_x=batch.nbr_edge_x[h],
        )  # fmt: skip
        for key, t in got.items():
            name = f'b{b}_h{h}_{key}'
            val = t.cpu().numpy()
            if meta.get('digest_only'):
                assert hashlib.sha256(np.ascontiguousarray(val).tobytes()).hexdigest() == meta['digests'][name], f'{tag} {name}'
            else:
                exp = a[name]
                assert val.dtype == exp.dtype, f'{tag} {name}: dtype {val.dtype} vs {exp.dtype}'
                assert val.shape == exp.shape, f'{tag} {name}: shape {val.shape} vs {exp.shape}'
                np.testing.assert_array_equal(val, exp, err_msg=f'{tag} {name}')


def _edge_x_of(case, a):
    if case == 'g3_wiki_medium':
        from tgm_amd.synth import make_stream

        return make_stream('wiki', seed=1337, num_edges=20_000, edge_dim=8).edge_x
    return torch.from_numpy(a['edge_x']) if 'edge_x' in a else None


@pytest.mark.parametrize('case', gu.sampler_cases() + ['g3_wiki_medium'])
def test_ring_mode_matches_reference_goldens(case):
    """Default (streaming) mode == the reference on every fixture, incl. the int32-key regime (g3)."""
    _, DGDataLoader, _, HookManager, RecencyNeighborHook, ReplayNegatives = _mk()
    meta, a = gu.load(case)
    edge_x = _edge_x_of(case, a)
    keys = ['edge_src', 'edge_dst'] + (['neg'] if meta['has_neg'] else [])
    tkeys = ['edge_time', 'edge_time'] + (['neg_time'] if meta['has_neg'] else [])
    hook = RecencyNeighborHook(meta['num_nodes'], meta['num_nbrs'], keys, tkeys, directed=meta['directed'])
    hm = HookManager(keys=['k'])
    replay = None
    if meta['has_neg']:
        replay = ReplayNegatives(torch.from_numpy(a['neg']).to(DEV))
        hm.register('k', replay)
    hm.register('k', hook)
    graphs = [(lo, _graph(a, lo, hi, edge_x)) for lo, hi in meta['segments']]
    b = 0
    with hm.activate('k'):
        for step in meta['plan']:
            if step == 'reset':
                hm.reset_state()
                continue
            for lo, dg in graphs:
                if replay is not None:
                    replay.base = lo
                for batch in DGDataLoader(dg, batch_size=meta['batch_size'], hook_manager=hm):
                    _compare(meta, a, b, batch, len(meta['num_nbrs']), case)
                    b += 1
    assert b == meta['num_batches']


@pytest.mark.parametrize('case', gu.sampler_cases())
def test_csr_mode_matches_reference_goldens(case):
    """Static-index mode on one resident store with train/val as views of it."""
    DGData, DGDataLoader, DGraph, HookManager, RecencyNeighborHook, ReplayNegatives = _mk()
    meta, a = gu.load(case)
    edge_x = _edge_x_of(case, a)
    E = len(a['src'])
    full = _graph(a, 0, E, edge_x)
    keys = ['edge_src', 'edge_dst'] + (['neg'] if meta['has_neg'] else [])
    tkeys = ['edge_time', 'edge_time'] + (['neg_time'] if meta['has_neg'] else [])
    hook = RecencyNeighborHook(
        meta['num_nodes'], meta['num_nbrs'], keys, tkeys, directed=meta['directed'], mode='csr', batch_starts=gu.batch_starts(meta)
    )
    hm = HookManager(keys=['k'])
    if meta['has_neg']:
        hm.register('k', ReplayNegatives(torch.from_numpy(a['neg']).to(DEV)))
    hm.register('k', hook)
    b = 0
    with hm.activate('k'):
        for step in meta['plan']:
            if step == 'reset':
                hm.reset_state()
                continue
            for lo, hi in meta['segments']:
                view = full.slice_events(lo, hi)
                for batch in DGDataLoader(view, batch_size=meta['batch_size'], hook_manager=hm):
                    _compare(meta, a, b, batch, len(meta['num_nbrs']), case)
                    b += 1
    assert b == meta['num_batches']


def _random_stream(seed, N, E, D, tmax):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, N, E).astype(np.int32)
    dst = rng.integers(0, N, E).astype(np.int32)
    ts = np.sort(rng.integers(1, tmax, E)).astype(np.int64)
    x = rng.random((E, D), dtype=np.float32) if D else None
    neg = rng.integers(0, N, E).astype(np.int32)
    return dict(src=src, dst=dst, ts=ts, neg=neg), (None if x is None else torch.from_numpy(x))


@pytest.mark.parametrize(
    'N,E,D,tmax,num_nbrs,bs,directed,key_arith',
    [
        (300, 6000, 12, 50_000, [10, 10], 100, False, 'int32'),  # no wrap: 300*50k < 2^31
        (3000, 8000, 4, 3_000_000, [5, 4], 64, False, 'int32'),  # wrapping keys
        (3000, 8000, 4, 3_000_000, [5, 4], 64, False, 'int64'),
        (500, 5000, 3, 1000, [70], 128, True, 'int32'),  # B > 64: general (chunked) path
        (200, 4000, 6, 400, [3, 8, 2], 50, False, 'int32'),  # k < B on some hops, heavy ties
        (100, 3000, 0, 300, [4, 4], 37, False, 'int32'),  # no edge features
        (50, 3000, 5, 200, [20], 1000, False, 'int32'),  # runs longer than B inside one batch (mid-size single-workgroup update, m=2000)
        (60, 9000, 5, 300, [20], 2500, False, 'int32'),  # m=5000 entries: radix-sort update path
        (4000, 9000, 3, 2_000_000, [6, 2], 1500, False, 'int32'),  # mid-size update (m=3000) with wrapping keys
        (9000, 12800, 4, 2_600_000, [20, 3], 1600, False, 'int32'),  # 8-rank global wiki batch: m=3200, wrapping keys
        (4000, 15000, 3, 2_000_000, [6, 2], 2500, False, 'int32'),  # m=5000: radix-sort path with wrapping keys
        (30, 24000, 2, 500, [8], 6000, False, 'int32'),  # m=12000: radix-sort path, hub runs far longer than B, heavy ties
        (400, 3000, 7, 2000, [6], 60, False, 'int32'),  # D not a multiple of 4 (scalar gather path)
        (400, 3000, 6, 2000, [6], 60, False, 'int32'),  # D % 2 == 0 (float2 path)
        (150, 4000, 4, 600, [64], 80, False, 'int32'),  # k = B = 64: the widest single-wave window
        (300, 2000, 5, 900, [1, 1, 1], 40, False, 'int32'),  # k = 1 on three hops
        (120, 3000, 8, 500, [32, 2], 64, True, 'int32'),  # B = 32: packed kernel group boundary, directed
    ],
)
def test_ring_mode_matches_oracle_random(N, E, D, tmax, num_nbrs, bs, directed, key_arith):
    _check_ring_against_oracle(N, E, D, tmax, num_nbrs, bs, directed, key_arith)


@pytest.mark.parametrize('validate', ['deferred', 'off', 'sync'])
@pytest.mark.parametrize(
    'N,E,D,tmax,num_nbrs,bs',
    [
        (3000, 6000, 4, 3_000_000, [5, 4], 64),  # wrapping keys, one-workgroup plan (m=128)
        (9000, 9600, 4, 2_600_000, [20, 3], 1600),  # 8-rank global wiki batch (m=3200: 4 elements per thread)
        (5000, 8000, 4, 2_600_000, [10, 2], 800),  # m=1600: 2 elements per thread
        (50, 3000, 5, 200, [20], 700),  # hub runs longer than B (m=1400)
        (2000, 4096, 4, 2_600_000, [8, 3], 512),  # m=1024: the largest batch whose placement rides with hop 1
        (2000, 4104, 4, 2_600_000, [8, 3], 513),  # m=1026: merge riders + placement launch
        (6000, 8192, 2, 2_600_000, [4, 2], 2048),  # m=4096: the largest batch of the chunked sort
        (6000, 8196, 2, 2_600_000, [4, 2], 2049),  # m=4098: radix-sort path
        (700, 3000, 6, 900, [12], 300),  # single hop: chunk sort rides hop 0, the merge is its own launch
        (700, 3000, 0, 900, [12, 4], 300),  # no edge features: the commit launch still writes records and write_pos
        (3000, 6000, 64, 2_600_000, [20, 20], 400),  # wide rows, m=800: fused hop 0 + 1 launch, a rider per chunk, the last one out places
        (500, 4626, 172, 3000, [20, 20], 257),  # m=514: three chunks (the last with two entries), heavy ties, ragged last batch
        (800, 4096, 32, 2_600_000, [16, 8], 512),  # m=1024 in the fused launch: four full chunks
    ],
)
def test_ring_step_variants(N, E, D, tmax, num_nbrs, bs, validate):
    """tgmx_recency_step in every validation mode (one call per batch, or lookups / check / update)."""
    _check_ring_against_oracle(N, E, D, tmax, num_nbrs, bs, False, 'int32', validate=validate)


@pytest.mark.skipif(any(os.environ.get(k) for k in ('TGMX_NO_RIDE', 'TGMX_NO_FUSE')), reason='the A/B knob removes the riders')
@pytest.mark.parametrize('D', [64, 4])
def test_riders_in_a_launch_many_times_larger_than_residency(D):
    """The riders' barrier (include/tgm_amd.h, "Riders and workgroup dispatch order") assumes that the riders -- the FIRST <= 16
    workgroups of a lookup launch -- are resident together, which in-order workgroup dispatch guarantees however large the grid is.
    Here the launch that carries them is 7 x what the chip holds at once: 60 000 hop-1 seeds (15 000 four-wave workgroups against
    ~2 048 resident) behind an m = 2 000 update (eight chunk riders, their barrier, the merge), wide rows (hops 0 + 1 fused) and narrow
    ones (hop 0's launch carries them).  Bit-exact against the oracle on every batch, no status bit, no timeout."""
    _check_ring_against_oracle(20_000, 12_000, D, 2_600_000, [20, 4], 1000, False, 'int32', validate='deferred')


def _check_ring_against_oracle(N, E, D, tmax, num_nbrs, bs, directed, key_arith, **hook_kw):
    from oracle.ring_port import RingSamplerCPU

    _, DGDataLoader, _, HookManager, RecencyNeighborHook, ReplayNegatives = _mk()
    a, edge_x = _random_stream(1234 + N + E, N, E, D, tmax)
    hook = RecencyNeighborHook(N, num_nbrs, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'],
                               directed=directed, key_arith=key_arith, **hook_kw)  # fmt: skip
    hm = HookManager(keys=['k'])
    hm.register('k', ReplayNegatives(torch.from_numpy(a['neg']).to(DEV)))
    hm.register('k', hook)
    oracle = RingSamplerCPU(N, num_nbrs, D, directed, key_arith=key_arith)
    dg = _graph(a, 0, E, edge_x)
    T = torch.from_numpy
    with hm.activate('k'):
        for b, batch in enumerate(DGDataLoader(dg, batch_size=bs, hook_manager=hm)):
            lo, hi = b * bs, min((b + 1) * bs, E)
            seeds = T(np.concatenate([a['src'][lo:hi], a['dst'][lo:hi], a['neg'][lo:hi]]))
            times = T(np.concatenate([a['ts'][lo:hi]] * 3))
            hops = oracle.step(seeds, times, T(a['src'][lo:hi]), T(a['dst'][lo:hi]), T(a['ts'][lo:hi]),
                               None if edge_x is None else edge_x[lo:hi])  # fmt: skip
            for h, (_, _, o_i, o_t, o_x) in enumerate(hops):
                assert torch.equal(batch.nbr_nids[h].cpu(), o_i), f'b{b} h{h} ids'
                assert torch.equal(batch.nbr_edge_time[h].cpu(), o_t), f'b{b} h{h} times'
                assert torch.equal(batch.nbr_edge_x[h].cpu(), o_x), f'b{b} h{h} feats'
            assert torch.equal(batch.seed_nids[0].cpu(), seeds) and torch.equal(batch.seed_times[0].cpu(), times)
    hook.check()


@pytest.mark.parametrize('bs,directed,tmax', [(129, False, 40_000), (300, False, 300), (511, False, 5), (512, False, 40_000), (257, True, 300),
                                              (1024, True, 40_000)])
def test_riders_spread_over_both_lookup_launches_vs_oracle(bs, directed, tmax):
    """Two packed lookup launches (narrow rows, two hops) with 256 < m <= 1024 update entries -- the review-shaped step: the chunk sorts ride
    hop 0, the merge (a rider per chunk) and the last-one-out placement ride hop 1 (round 6).  m = 257 ... 1024 (2 to 4 chunks, ragged last
    chunk), heavy ties (tmax = 5: the key order decides), ids / times / feature rows of every batch bit-exact against the oracle."""
    _check_ring_against_oracle(600, 7 * bs, 16, tmax, [10, 10], bs, directed, 'int32', validate='deferred')


def test_csr_mode_equals_ring_int64_random():
    """Stateless CSR lookup == streaming rings with the intended key order, mid-size, non-bipartite."""
    DGData, DGDataLoader, DGraph, HookManager, RecencyNeighborHook, ReplayNegatives = _mk()
    N, E, D, bs, num_nbrs = 2000, 40_000, 16, 200, [20, 20]
    a, edge_x = _random_stream(77, N, E, D, 20_000)  # ~2 edges per timestamp: ties across roles
    outs = {}
    for mode in ('ring', 'csr'):
        hook = RecencyNeighborHook(N, num_nbrs, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'],
                                   mode=mode, key_arith='int64', batch_size=bs, validate='deferred')  # fmt: skip
        hm = HookManager(keys=['k'])
        hm.register('k', ReplayNegatives(torch.from_numpy(a['neg']).to(DEV)))
        hm.register('k', hook)
        dg = _graph(a, 0, E, edge_x)
        res = []
        with hm.activate('k'):
            for batch in DGDataLoader(dg, batch_size=bs, hook_manager=hm):
                res.append([(batch.nbr_nids[h], batch.nbr_edge_time[h], batch.nbr_edge_x[h]) for h in range(2)])
        hook.check()
        outs[mode] = res
    assert len(outs['ring']) == len(outs['csr']) == E // bs
    for b, (r, c) in enumerate(zip(outs['ring'], outs['csr'])):
        for h in range(2):
            for x, y in zip(r[h], c[h]):
                assert torch.equal(x, y), f'batch {b} hop {h}'


def test_full_size_wiki_properties():
    """BASELINE config 2 at FULL size (N=9227, E=157474, D=172, bs=200, k=[20,20], 788 batches), size-independent
    properties checked on every batch: the streaming rings (intended key order) and the stateless index agree bit for
    bit; every row is oldest -> newest, strictly before its query time, pads left-aligned as (-1, 0, 0.0); every valid
    slot's feature row is a verbatim row of edge_x (found through its stream position); hop h+1 seeds are hop h outputs."""
    DGData, DGDataLoader, DGraph, HookManager, RecencyNeighborHook, ReplayNegatives = _mk()
    from tgm_amd.synth import make_stream

    st = make_stream('wiki', seed=1337)
    N, E, D, bs, ks = st.num_nodes, st.num_edges, st.edge_dim, 200, [20, 20]
    g = torch.Generator().manual_seed(5)
    neg = torch.randint(8227, N, (E,), generator=g, dtype=torch.int32)
    data = DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x)
    edge_x_dev = st.edge_x.to(DEV)
    ts_dev = st.ts.to(DEV)
    iters = {}
    for mode in ('ring', 'csr'):
        hook = RecencyNeighborHook(N, ks, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], mode=mode,
                                   key_arith='int64', batch_size=bs, validate='deferred')  # fmt: skip
        hm = HookManager(keys=['k'])
        hm.register('k', ReplayNegatives(neg.to(DEV)))
        hm.register('k', hook)
        iters[mode] = (hm, hook, DGDataLoader(DGraph(data, device=DEV), batch_size=bs, hook_manager=hm))
    (hm_r, hook_r, ld_r), (hm_c, hook_c, ld_c) = iters['ring'], iters['csr']
    nb = 0
    with hm_r.activate('k'), hm_c.activate('k'):
        for br, bc in zip(ld_r, ld_c):
            for h, k in enumerate(ks):
                n, t, x = br.nbr_nids[h], br.nbr_edge_time[h], br.nbr_edge_x[h]
                assert torch.equal(n, bc.nbr_nids[h]) and torch.equal(t, bc.nbr_edge_time[h]) and torch.equal(x, bc.nbr_edge_x[h]), (nb, h)
                valid = n >= 0
                q = br.seed_times[h][:, None]
                assert bool(((t < q) | ~valid).all()), 'a sampled edge is not strictly before its query time'
                assert bool((valid[:, 1:] | ~valid[:, :-1]).all()), 'pads must be left-aligned'
                assert bool(((t[:, 1:] >= t[:, :-1]) | ~valid[:, :-1]).all()), 'rows must run oldest -> newest'
                assert bool((t[~valid] == 0).all()) and bool((x[~valid] == 0).all())
                if h + 1 < len(ks):
                    assert torch.equal(br.seed_nids[h + 1], n.reshape(-1)) and torch.equal(br.seed_times[h + 1], t.reshape(-1))
            if nb % 97 == 0:  # feature rows are verbatim copies: locate each valid slot's edge by (seed, nbr, time)
                n, t, x = br.nbr_nids[0], br.nbr_edge_time[0], br.nbr_edge_x[0]
                seeds = br.seed_nids[0]
                rows = torch.nonzero(n >= 0)[:200]
                for r, c in rows.tolist():
                    a_, b_, tt = int(seeds[r]), int(n[r, c]), int(t[r, c])
                    lo, hi = int(torch.searchsorted(ts_dev, tt)), int(torch.searchsorted(ts_dev, tt, right=True))
                    s_, d_ = st.src[lo:hi], st.dst[lo:hi]
                    cand = torch.nonzero(((s_ == a_) & (d_ == b_)) | ((s_ == b_) & (d_ == a_))).reshape(-1) + lo
                    assert any(torch.equal(edge_x_dev[e], x[r, c]) for e in cand.tolist()), (nb, r, c)
            nb += 1
    hook_r.check()
    hook_c.check()
    assert nb == (E + bs - 1) // bs


def test_csr_mode_with_irregular_batch_starts():
    """Static index built from explicit, irregular batch boundaries (time-unit batching gives such schedules) against
    the streaming rings fed the same batches; also a directed stream."""
    DGData, DGDataLoader, DGraph, HookManager, RecencyNeighborHook, ReplayNegatives = _mk()
    N, E, D, ks = 300, 5000, 4, [7, 3]
    a, edge_x = _random_stream(5, N, E, D, 2500)  # ~2 edges per timestamp
    rng = np.random.default_rng(9)
    cuts = np.unique(np.concatenate([[0], np.sort(rng.integers(1, E, 60)), [E]]))
    starts = cuts[:-1].tolist()
    for directed in (False, True):
        hooks = {
            'ring': RecencyNeighborHook(N, ks, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], mode='ring', key_arith='int64', directed=directed),
            'csr': RecencyNeighborHook(N, ks, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], mode='csr', batch_starts=starts, directed=directed),
        }
        dg = _graph(a, 0, E, edge_x)
        neg = torch.from_numpy(a['neg']).to(DEV)
        for lo, hi in zip(cuts[:-1].tolist(), cuts[1:].tolist()):
            outs = {}
            for mode, hook in hooks.items():
                batch = dg.slice_events(lo, hi).materialize()
                batch.neg, batch.neg_time = neg[lo:hi].clone(), batch.edge_time.clone()
                b = hook(dg, batch)
                outs[mode] = [(b.nbr_nids[h], b.nbr_edge_time[h], b.nbr_edge_x[h]) for h in range(2)]
            for h in range(2):
                for x, y in zip(outs['ring'][h], outs['csr'][h]):
                    assert torch.equal(x, y), (directed, lo, hi, h)


def test_direct_entry_points_match_the_step_call():
    """tgmx_ring_lookup / tgmx_ring_update / tgmx_recency_lookup_csr called one by one (what INTEGRATION.md's ctypes stub
    does) give exactly what the hooks get from tgmx_recency_step."""
    DGData, DGDataLoader, DGraph, HookManager, RecencyNeighborHook, ReplayNegatives = _mk()
    from tgm_amd import _native
    from tgm_amd.index import build_csr

    lib = _native.load()
    N, E, D, bs, ks = 500, 6000, 8, 150, [6, 4]
    a, edge_x = _random_stream(31, N, E, D, 4000)
    B = max(ks)
    # reference path: the hooks (one step call per batch)
    outs = {}
    for mode in ('ring', 'csr'):
        hook = RecencyNeighborHook(N, ks, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], mode=mode, key_arith='int64', batch_size=bs)
        hm = HookManager(keys=['k'])
        hm.register('k', ReplayNegatives(torch.from_numpy(a['neg']).to(DEV)))
        hm.register('k', hook)
        with hm.activate('k'):
            outs[mode] = [[(b.nbr_nids[h].clone(), b.nbr_edge_time[h].clone(), b.nbr_edge_x[h].clone()) for h in range(2)]
                          for b in DGDataLoader(_graph(a, 0, E, edge_x), batch_size=bs, hook_manager=hm)]  # fmt: skip
    # direct path
    T = lambda x: torch.from_numpy(x).to(DEV)
    src, dst, ts, neg, ex = T(a['src']), T(a['dst']), T(a['ts']), T(a['neg']), edge_x.to(DEV)
    ring = torch.empty((N * B, 2), dtype=torch.int64, device=DEV)
    wpos = torch.empty(N, dtype=torch.int32, device=DEV)
    ring_x = torch.empty((N * B, D), dtype=torch.float32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    st = _native.stream_ptr(0)
    assert lib.tgmx_ring_reset(ring.data_ptr(), wpos.data_ptr(), B, N, st) == 0
    scratch = torch.zeros(int(lib.tgmx_ring_update_scratch_bytes(bs, 0)), dtype=torch.uint8, device=DEV)
    csr = build_csr(src, dst, ts, N, batch_size=bs)
    for b in range(E // bs):
        lo, hi = b * bs, (b + 1) * bs
        seeds = torch.cat([src[lo:hi], dst[lo:hi], neg[lo:hi]])
        times = torch.cat([ts[lo:hi]] * 3)
        for mode in ('ring', 'csr'):
            cur_n, cur_t = seeds, times
            for h, k in enumerate(ks):
                S = cur_n.numel()
                nid = torch.empty((S, k), dtype=torch.int32, device=DEV)
                nts = torch.empty((S, k), dtype=torch.int64, device=DEV)
                nx = torch.empty((S, k, D), dtype=torch.float32, device=DEV)
                if mode == 'ring':
                    rc = lib.tgmx_ring_lookup(ring.data_ptr(), wpos.data_ptr(), ring_x.data_ptr(), D, cur_n.data_ptr(), cur_t.data_ptr(), S, k, B, N,
                                              1 if h else 0, nid.data_ptr(), nts.data_ptr(), nx.data_ptr(), status.data_ptr(), st, None, None)
                else:
                    rc = lib.tgmx_recency_lookup_csr(csr.indptr.data_ptr(), csr.adj.data_ptr(), ex.data_ptr(), D, cur_n.data_ptr(), cur_t.data_ptr(), S, k,
                                                     B, 0, lo, N, 1 if h else 0, nid.data_ptr(), nts.data_ptr(), nx.data_ptr(), status.data_ptr(), st, None, None)
                assert rc == 0
                r_n, r_t, r_x = outs[mode][b][h]
                assert torch.equal(nid, r_n) and torch.equal(nts, r_t) and torch.equal(nx, r_x), (mode, b, h)
                cur_n, cur_t = nid.view(-1), nts.view(-1)
        rc = lib.tgmx_ring_update(ring.data_ptr(), wpos.data_ptr(), ring_x.data_ptr(), D, B, N, src[lo:hi].data_ptr(), dst[lo:hi].data_ptr(),
                                  ts[lo:hi].data_ptr(), ex[lo:hi].data_ptr(), bs, lo, 0, 0, scratch.data_ptr(), status.data_ptr(), st)
        assert rc == 0
    assert int(status.item()) == 0
    # timing events of the ABI
    t = _native.KernelTimer()
    S = 64
    nid = torch.empty((S, 4), dtype=torch.int32, device=DEV); nts = torch.empty((S, 4), dtype=torch.int64, device=DEV); nx = torch.empty((S, 4, D), device=DEV)
    assert lib.tgmx_ring_lookup(ring.data_ptr(), wpos.data_ptr(), ring_x.data_ptr(), D, seeds.data_ptr(), times.data_ptr(), S, 4, B, N, 0, nid.data_ptr(),
                                nts.data_ptr(), nx.data_ptr(), status.data_ptr(), st, t.start, t.stop) == 0
    torch.cuda.synchronize()
    assert 0.0 < t.elapsed_ms() < 50.0


def test_edge_cases_and_errors():
    DGData, DGDataLoader, DGraph, HookManager, RecencyNeighborHook, _ = _mk()
    a = dict(src=np.array([1, 2, 3], np.int32), dst=np.array([2, 3, 4], np.int32), ts=np.array([1, 2, 3], np.int64))
    dg = _graph(a, 0, 3, None)
    # no edge features -> [S, k, 0]
    hook = RecencyNeighborHook(5, [1], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time'], directed=True)
    batch = hook(dg, dg.materialize())
    assert batch.nbr_edge_x[0].shape == (6, 1, 0) and batch.nbr_nids[0].shape == (6, 1)
    assert batch.seed_node_nbr_mask['edge_dst'].tolist() == [3, 4, 5]
    # id suffix
    hook = RecencyNeighborHook(5, [1], ['edge_src'], ['edge_time'], id='foo')
    batch = hook(dg, dg.materialize())
    assert hasattr(batch, 'nbr_nids_foo') and hasattr(batch, 'seed_node_nbr_mask_foo')
    # out-of-range seed id / negative time raise ValueError (kernel-side validation, sync mode)
    hook = RecencyNeighborHook(5, [1], ['foo'], ['bar'])
    batch = dg.materialize()
    batch.foo, batch.bar = torch.tensor([7], dtype=torch.int32, device=DEV), torch.tensor([1], device=DEV)
    with pytest.raises(ValueError):
        hook(dg, batch)
    batch.foo = torch.tensor([-1], dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError):
        hook(dg, batch)
    batch.foo, batch.bar = torch.tensor([1], dtype=torch.int32, device=DEV), torch.tensor([-4], device=DEV)
    with pytest.raises(ValueError):
        hook(dg, batch)
    # structural errors
    batch.foo = 'nope'
    with pytest.raises(ValueError):
        hook(dg, batch)
    batch.foo = torch.zeros(2, 3, device=DEV)
    batch.bar = torch.zeros(2, 3, device=DEV)
    with pytest.raises(ValueError):
        hook(dg, batch)
    del batch.foo
    with pytest.raises(ValueError):
        hook(dg, batch)
    # None seeds: warn once, emit empties, skip the update
    batch.foo, batch.bar = None, None
    with pytest.warns(UserWarning):
        out = hook(dg, batch)
    assert out.nbr_nids[0].numel() == 0 and out.nbr_edge_x[0].shape == (0, 0)


@pytest.mark.skipif(any(os.environ.get(k) for k in ('TGMX_NO_RIDE', 'TGMX_NO_FUSE')), reason='the A/B knob removes the barrier')
def test_dirty_scratch_head_is_reported_not_hung():
    """The riders' barrier lives in the first 256 bytes of the update scratch, which must be zero at first use.  A
    dirty head must not hang the device: the barrier gives up after about a second and raises TGMX_ST_SCRATCH."""
    DGData, DGDataLoader, DGraph, HookManager, RecencyNeighborHook, ReplayNegatives = _mk()
    N, E, D, bs = 3000, 4000, 64, 800  # m = 1600, wide rows: hops fused, riders = sort | barrier | merge
    a, edge_x = _random_stream(5, N, E, D, 100_000)
    hook = RecencyNeighborHook(N, [4, 20], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time'], validate='deferred')
    hm = HookManager(keys=['k'])
    hm.register('k', hook)
    dg = _graph(a, 0, E, edge_x)
    loader = DGDataLoader(dg, batch_size=bs, hook_manager=hm, output_pool=0)  # hook by hook: the hook's own argument block is the one inspected
    with hm.activate('k'):
        it = iter(loader)
        next(it)  # allocates (and zeroes) the scratch
        hook.check()
        assert hook.fuses_first_hops()
        hook._scratch[:8].fill_(0x7F)  # corrupt the barrier words
        next(it)
        with pytest.raises(RuntimeError, match='scratch'):
            hook.check()


def test_violated_timestamp_bound_is_reported():
    """Large batches sort only the key bits that `ts_bound` allows; a timestamp beyond the promise must be reported."""
    DGData, DGDataLoader, DGraph, HookManager, RecencyNeighborHook, ReplayNegatives = _mk()
    N, E, D, bs = 500, 6000, 2, 3000  # m = 6000: the rocPRIM path
    a, edge_x = _random_stream(9, N, E, D, 50_000)
    hook = RecencyNeighborHook(N, [4], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time'], validate='deferred')
    hm = HookManager(keys=['k'])
    hm.register('k', hook)
    loader = DGDataLoader(_graph(a, 0, E, edge_x), batch_size=bs, hook_manager=hm, output_pool=0)  # hook by hook: the hook's own argument block
    with hm.activate('k'):
        it = iter(loader)
        next(it)
        hook.check()  # the bound taken from the store holds
        assert hook._step.ts_bound == int(a['ts'][-1]) == hook._store_promise[0]
        hook._store_promise = (10, hook._store_promise[1])  # a promise the second batch breaks
        next(it)
        with pytest.raises(RuntimeError, match='ts_bound'):
            hook.check()


def _fuzz_config(seed):
    """One random sampler configuration per seed: sizes chosen to land on every update plan (one workgroup, chunk sort +
    merge, radix sort), every lookup specialisation (packed groups, one wave, chunked B > 64) and both key arithmetics."""
    r = np.random.default_rng(10_000 + seed)
    N = int(r.choice([7, 40, 300, 2500, 9000]))
    bs = int(r.choice([1, 17, 64, 200, 513, 900, 1600, 2300]))
    E = int(min(max(bs * r.integers(3, 9), 200), 14_000))
    D = int(r.choice([0, 1, 2, 3, 4, 6, 8, 16, 33]))
    tmax = int(r.choice([5, 300, 40_000, 3_000_000]))
    L = int(r.integers(1, 4))
    num_nbrs = [int(r.choice([1, 2, 3, 5, 8, 16, 20, 32, 64, 70])) for _ in range(L)]
    while int(np.prod(num_nbrs)) * 3 * bs > 400_000:  # keep the deepest hop small enough for the CPU oracle
        num_nbrs[int(np.argmax(num_nbrs))] = max(1, max(num_nbrs) // 2)
    directed = bool(r.integers(0, 2))
    key_arith = str(r.choice(['int32', 'int64']))
    validate = str(r.choice(['deferred', 'off', 'sync']))
    return N, E, D, tmax, num_nbrs, bs, directed, key_arith, validate


@pytest.mark.parametrize('seed', range(int(os.environ.get('TGMX_FUZZ', '16'))))
def test_ring_mode_fuzz(seed):
    """Seeded random configurations against the oracle (TGMX_FUZZ=<n> widens the sweep; the default keeps the suite short)."""
    N, E, D, tmax, num_nbrs, bs, directed, key_arith, validate = _fuzz_config(seed)
    _check_ring_against_oracle(N, E, D, tmax, num_nbrs, bs, directed, key_arith, validate=validate)


@pytest.mark.parametrize('seed', range(max(4, int(os.environ.get('TGMX_FUZZ', '16')) // 2)))
def test_csr_mode_fuzz(seed):
    """Stateless static-index lookups == streaming rings with the intended (int64) key order, on seeded random
    configurations (wide and narrow rows, so both the fused hop 0 + hop 1 launch and the per-hop launches run)."""
    DGData, DGDataLoader, DGraph, HookManager, RecencyNeighborHook, ReplayNegatives = _mk()
    r = np.random.default_rng(20_000 + seed)
    N = int(r.choice([9, 120, 2500]))
    bs = int(r.choice([7, 64, 200, 700]))
    E = int(min(max(bs * r.integers(4, 10), 300), 6000))
    D = int(r.choice([0, 3, 8, 64, 172]))
    tmax = int(r.choice([20, 5000, 2_000_000]))
    L = int(r.integers(1, 4))
    num_nbrs = [int(r.choice([1, 3, 8, 20, 32])) for _ in range(L)]
    while int(np.prod(num_nbrs)) * 3 * bs * max(D, 1) > 40_000_000:
        num_nbrs[int(np.argmax(num_nbrs))] = max(1, max(num_nbrs) // 2)
    directed = bool(r.integers(0, 2))
    a, edge_x = _random_stream(4321 + seed, N, E, D, tmax)
    outs = {}
    for mode in ('ring', 'csr'):
        hook = RecencyNeighborHook(N, num_nbrs, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'],
                                   mode=mode, key_arith='int64', batch_size=bs, directed=directed, validate='deferred')  # fmt: skip
        hm = HookManager(keys=['k'])
        hm.register('k', ReplayNegatives(torch.from_numpy(a['neg']).to(DEV)))
        hm.register('k', hook)
        dg = _graph(a, 0, E, edge_x)
        res = []
        with hm.activate('k'):
            for batch in DGDataLoader(dg, batch_size=bs, hook_manager=hm):
                res.append([(batch.nbr_nids[h], batch.nbr_edge_time[h], batch.nbr_edge_x[h]) for h in range(L)])
        hook.check()
        outs[mode] = res
    assert len(outs['ring']) == len(outs['csr'])
    for b, (x, y) in enumerate(zip(outs['ring'], outs['csr'])):
        for h in range(L):
            for u, v in zip(x[h], y[h]):
                assert torch.equal(u, v), f'batch {b} hop {h} ({N=}, {bs=}, {D=}, {num_nbrs=}, {directed=})'


# ---- the BASELINE.json configurations at their own shapes (close the gaps) -----------------------------------
def _pooled_pipeline(st, bs, ks, mode='ring', pool=3, key_arith='int32', neg_seed=3, dg=None, edge_features='dense'):
    """negatives -> recency sampler through DGDataLoader(output_pool=): the path bench.py times."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook

    if dg is None:
        dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', RandomNegativeEdgeSamplerHook(int(st.dst.min()), st.num_nodes, seed=neg_seed))
    hook = RecencyNeighborHook(st.num_nodes, ks, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], mode=mode,
                               key_arith=key_arith, validate='deferred', batch_size=bs if mode == 'csr' else None, edge_features=edge_features)  # fmt: skip
    hm.register('k', hook)
    return dg, hm, hook, DGDataLoader(dg, batch_size=bs, hook_manager=hm, output_pool=pool)


def _against_oracle(st, bs, ks, n_batches, pool=3):
    from oracle.ring_port import RingSamplerCPU

    _, hm, hook, loader = _pooled_pipeline(st, bs, ks, pool=pool)
    ref = RingSamplerCPU(st.num_nodes, ks, st.edge_dim)
    src, dst, ts, x = st.src.cpu(), st.dst.cpu(), st.ts.cpu(), st.edge_x.cpu()
    with hm.activate('k'):
        for b, batch in enumerate(loader):
            if b == n_batches:
                break
            lo, hi = b * bs, min((b + 1) * bs, st.num_edges)
            hops = ref.step(torch.cat([src[lo:hi], dst[lo:hi], batch.neg.cpu()]), torch.cat([ts[lo:hi]] * 3), src[lo:hi], dst[lo:hi], ts[lo:hi], x[lo:hi])
            for h, (_, _, o_i, o_t, o_x) in enumerate(hops):
                assert torch.equal(batch.nbr_nids[h].cpu(), o_i), f'b{b} h{h} ids'
                assert torch.equal(batch.nbr_edge_time[h].cpu(), o_t), f'b{b} h{h} times'
                assert torch.equal(batch.nbr_edge_x[h].cpu(), o_x), f'b{b} h{h} feats'
    hook.check()
    assert loader._compiled[1] is not None


def test_cfg2_benched_mode_full_size_vs_oracle():
    """BASELINE cfg 2 exactly as bench.py runs it -- full N = 9227, D = 172, bs = 200, k = [20, 20], the reference's
    wrapping int32 key arithmetic (recency.py:347), pooled outputs, negatives generated in the seed fetch -- against the
    CPU restatement of the reference for the first 90 batches (18 000 edges: rings of the hubs wrap several times).
    Three pipelines over the same stream in lockstep against ONE pass of the oracle (the CPU restatement is what this test's minutes
    are: it used to run once per pool setting): pool = 1, the benched pool (ONE persistent output set, delta feature writes at D = 172);
    pool = 3; pool = None, the loader's default (liveness-checked sets: the loop below holds batch i while batch i + 1 is produced,
    so two sets alternate)."""
    from oracle.ring_port import RingSamplerCPU
    from tgm_amd.synth import make_stream

    st, bs, ks, n_batches = make_stream('wiki', seed=1337), 200, [20, 20], 90
    pipes = {pool: _pooled_pipeline(st, bs, ks, pool=pool) for pool in (1, 3, None)}
    ref = RingSamplerCPU(st.num_nodes, ks, st.edge_dim)
    src, dst, ts, x = st.src.cpu(), st.dst.cpu(), st.ts.cpu(), st.edge_x.cpu()
    (_, hm1, _, ld1), (_, hm3, _, ld3), (_, hmn, _, ldn) = pipes[1], pipes[3], pipes[None]
    with hm1.activate('k'), hm3.activate('k'), hmn.activate('k'):
        for b, batches in enumerate(zip(ld1, ld3, ldn)):
            if b == n_batches:
                break
            neg = batches[0].neg.cpu()
            lo, hi = b * bs, min((b + 1) * bs, st.num_edges)
            hops = ref.step(torch.cat([src[lo:hi], dst[lo:hi], neg]), torch.cat([ts[lo:hi]] * 3), src[lo:hi], dst[lo:hi], ts[lo:hi], x[lo:hi])
            for pool, batch in zip((1, 3, None), batches):
                assert torch.equal(batch.neg.cpu(), neg), f'b{b} pool {pool}: the generated negatives do not depend on the pool'
                for h, (_, _, o_i, o_t, o_x) in enumerate(hops):
                    assert torch.equal(batch.nbr_nids[h].cpu(), o_i), f'b{b} h{h} ids (pool {pool})'
                    assert torch.equal(batch.nbr_edge_time[h].cpu(), o_t), f'b{b} h{h} times (pool {pool})'
                    assert torch.equal(batch.nbr_edge_x[h].cpu(), o_x), f'b{b} h{h} feats (pool {pool})'
    for pool, (_, _, hook, loader) in pipes.items():
        hook.check()
        assert loader._compiled[1] is not None


def test_cfg3_review_shape_two_hops_vs_oracle():
    """BASELINE cfg 3's sampler at its own shape: bipartite, unix-scale timestamps, D = 16, bs = 512, k = [10, 10]
    (hop 1 runs the packed narrow-row kernel with 16-lane groups, the update's placement rides as m = 1024)."""
    from tgm_amd.synth import make_stream

    _against_oracle(make_stream('review', seed=21, num_edges=30_000, n_src=30_000, n_dst=5_000), 512, [10, 10], 1000)


def test_cfg3_review_full_size_midstream_properties():
    """BASELINE cfg 3's sampler at FULL size (N = 350 k, E = 4.8 M, D = 16, bs = 512, k = [10, 10]): an epoch opened mid-stream at
    edge 2.4 M and followed for 300 batches -- the streaming rings (hop 1 through the packed 16-lane-group kernel with the update's
    m = 1024 placement riding, dense copies AND edge ids) and the static index over the whole 4.8 M-edge store agree bit for bit on
    every batch, the published edge ids address exactly the copied feature rows, and every row satisfies the sampler's invariants
    (strictly earlier than the seed, right-aligned, oldest -> newest, pads zero)."""
    from tgm_amd import DGData, DGraph
    from tgm_amd.synth import make_stream

    st = make_stream('review', seed=1337, device=DEV)
    assert st.num_nodes > 300_000 and st.num_edges > 4_000_000
    bs, ks, first, nb = 512, [10, 10], 2_400_000 // 512 * 512, 300
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    view = dg.slice_events(first, first + nb * bs)
    _, hm_r, hook_r, ld_r = _pooled_pipeline(st, bs, ks, mode='ring', key_arith='int64', dg=view)
    _, hm_c, hook_c, ld_c = _pooled_pipeline(st, bs, ks, mode='csr', dg=view)
    _, hm_i, hook_i, ld_i = _pooled_pipeline(st, bs, ks, mode='ring', key_arith='int64', dg=view, edge_features='by_id')
    table = dg._storage.on(torch.device(DEV)).edge_x
    n_seen, n_valid = 0, 0
    with hm_r.activate('k'), hm_c.activate('k'), hm_i.activate('k'):
        for br, bc, bi in zip(ld_r, ld_c, ld_i):
            assert br._edge_lo == bc._edge_lo == bi._edge_lo == first + n_seen * bs
            assert torch.equal(br.neg, bc.neg) and torch.equal(br.neg, bi.neg)
            for h in range(2):
                n, t, x = br.nbr_nids[h], br.nbr_edge_time[h], br.nbr_edge_x[h]
                assert torch.equal(n, bc.nbr_nids[h]) and torch.equal(t, bc.nbr_edge_time[h]) and torch.equal(x, bc.nbr_edge_x[h]), (n_seen, h)
                assert torch.equal(n, bi.nbr_nids[h]) and torch.equal(t, bi.nbr_edge_time[h]), (n_seen, h, 'by id')
                valid = n >= 0
                eid = bi.nbr_edge_x.eids[h]
                assert bool((eid[~valid] == -1).all()) and bool((eid[valid] >= 0).all())
                assert torch.equal(table[eid[valid].long()], x[valid]), (n_seen, h, 'edge ids address the copied rows')
                q = br.seed_times[h][:, None]
                assert bool(((t < q) | ~valid).all()) and bool((valid[:, 1:] | ~valid[:, :-1]).all())
                assert bool(((t[:, 1:] >= t[:, :-1]) | ~valid[:, :-1]).all())
                assert bool((t[~valid] == 0).all()) and bool((x[~valid] == 0).all())
                if h == 1:
                    n_valid += int(valid.sum())
            assert torch.equal(br.seed_nids[1], br.nbr_nids[0].reshape(-1))
            n_seen += 1
    hook_r.check()
    hook_c.check()
    hook_i.check()
    assert n_seen == nb and n_valid > 100_000


def test_cfg4_comment_shape_reduced_vs_oracle():
    """BASELINE cfg 4 reduced in N and E only: non-bipartite (a node meets itself in both roles, timestamp ties inside a
    batch), unix-scale timestamps with wrapping int32 keys, D = 16, bs = 4096 (m = 8192: the rocPRIM radix-sort update
    with its front half on the side stream), k = [20, 20] (packed narrow-row kernel, 32-lane groups)."""
    from tgm_amd.synth import make_stream

    _against_oracle(make_stream('comment', seed=4, num_edges=9 * 4096 + 77, n_src=50_000), 4096, [20, 20], 1000)


@pytest.mark.parametrize('bs,ks,D,pool', [(1200, [10, 10], 16, 1), (1200, [10, 10], 6, 3), (1800, [20, 20], 16, 1), (1300, [10, 10], 4, 3), (1100, [12, 12], 8, 1),
                                          (1500, [20, 10], 16, 1), (11000, [10, 10], 16, 3)])
def test_tile_kernels_cooperative_index_phase_vs_oracle(bs, ks, D, pool):
    """The narrow-row TILE kernels at shapes the BASELINE configurations do not reach (hop 1 >= 2 tiles per CU, i.e. >= 32 768 seeds):
    B = 10 (six windows per load instruction of the cooperative index phase, lookup_tile_coop_kernel<.., 10, ..>), B = 12 (five per
    instruction through the BCAP = 20 body), B = 20; D = 16 (16-byte pieces), D = 4 (one piece per row), D = 6 / 8 (the float-piece copy);
    m = 2 bs <= 4096 entries, so the ring update's placement RIDES the tile launch (the RIDE instantiations); pool = 1: delta feature
    writes into one persistent set; k1 < B (rows narrower than the window); bs = 11 000: HOP 0 is a tile launch too (33 000 seeds drawn
    from the seed groups, negatives generated in the fetch) and the update takes the radix-sort path.  Rings against the CPU restatement
    of the reference bit for bit, the static index against the rings."""
    from tgm_amd.synth import make_stream

    st = make_stream('comment', seed=11, num_edges=(4 if bs > 4096 else 7) * bs + 19, n_src=6_000, edge_dim=D)
    _against_oracle(st, bs, ks, 1000, pool=pool)
    _, hm_r, hook_r, ld_r = _pooled_pipeline(st, bs, ks, mode='ring', key_arith='int64', pool=3)
    _, hm_c, hook_c, ld_c = _pooled_pipeline(st, bs, ks, mode='csr', pool=3)
    with hm_r.activate('k'), hm_c.activate('k'):
        for b, (br, bc) in enumerate(zip(ld_r, ld_c)):
            for h in range(2):
                assert torch.equal(br.nbr_nids[h], bc.nbr_nids[h]) and torch.equal(br.nbr_edge_time[h], bc.nbr_edge_time[h]), (b, h)
                assert torch.equal(br.nbr_edge_x[h], bc.nbr_edge_x[h]), (b, h, 'features')
    hook_r.check()
    hook_c.check()


def _tile_fuzz_seeds():
    import os

    return list(range(int(os.environ.get('TGMX_FUZZ', 0)) or 4))


@pytest.mark.parametrize('seed', _tile_fuzz_seeds())
def test_tile_kernels_fuzz_vs_oracle(seed):
    """Seeded random shapes in the tile kernels' range (hop 1 >= 32 768 seeds): B anywhere in 4..20 (windows shorter than their lane group,
    3 / 6 windows per load instruction), k1 <= k0 or k1 >= k0, D in {4, 6, 8, 12, 16}, pools of 1 and 3, few nodes (rings wrap, hubs, pad rows):
    rings against the CPU restatement of the reference bit for bit, the static index against the rings.  TGMX_FUZZ=n runs n configurations."""
    from tgm_amd.synth import make_stream

    rng = np.random.default_rng(9000 + seed)
    k0, k1 = int(rng.integers(4, 21)), int(rng.integers(4, 21))
    D = int(rng.choice([4, 6, 8, 12, 16]))
    bs = int(np.ceil(33_000 / (3 * k0))) + int(rng.integers(0, 400))
    if bs > 2048:  # (keep m = 2 bs on the riders' path here; the radix-sort path has its own case above)
        bs = 2048
        k0 = max(k0, int(np.ceil(33_000 / (3 * bs))))
    pool = int(rng.choice([1, 3]))
    st = make_stream('comment', seed=100 + seed, num_edges=5 * bs + int(rng.integers(1, 50)), n_src=int(rng.integers(800, 5000)), edge_dim=D)
    ks = [k0, k1]
    _against_oracle(st, bs, ks, 1000, pool=pool)
    _, hm_r, hook_r, ld_r = _pooled_pipeline(st, bs, ks, mode='ring', key_arith='int64', pool=3)
    _, hm_c, hook_c, ld_c = _pooled_pipeline(st, bs, ks, mode='csr', pool=3)
    with hm_r.activate('k'), hm_c.activate('k'):
        for b, (br, bc) in enumerate(zip(ld_r, ld_c)):
            for h in range(2):
                assert torch.equal(br.nbr_nids[h], bc.nbr_nids[h]) and torch.equal(br.nbr_edge_time[h], bc.nbr_edge_time[h]), (seed, ks, D, bs, b, h)
                assert torch.equal(br.nbr_edge_x[h], bc.nbr_edge_x[h]), (seed, ks, D, bs, b, h, 'features')
    hook_r.check()
    hook_c.check()


def test_cfg4_comment_full_size_midstream_properties():
    """BASELINE cfg 4 at FULL size (N = 1 M, E = 44 M, D = 16, bs = 4096, k = [20, 20]), an epoch opened mid-stream at
    edge 22 M and followed for 120 batches: the streaming rings (intended key order) and the static index over the
    whole 44 M-edge store agree bit for bit on every batch, and every row satisfies the sampler's invariants."""
    from tgm_amd import DGData, DGraph
    from tgm_amd.synth import make_stream

    st = make_stream('comment', seed=1337, device=DEV)
    bs, ks, first, nb = 4096, [20, 20], 22_000_000 // 4096 * 4096, 120
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device=DEV)
    view = dg.slice_events(first, first + nb * bs)
    _, hm_r, hook_r, ld_r = _pooled_pipeline(st, bs, ks, mode='ring', key_arith='int64', dg=view)
    _, hm_c, hook_c, ld_c = _pooled_pipeline(st, bs, ks, mode='csr', dg=view)
    n_seen, n_valid = 0, 0
    with hm_r.activate('k'), hm_c.activate('k'):
        for br, bc in zip(ld_r, ld_c):
            assert br._edge_lo == bc._edge_lo == first + n_seen * bs
            assert torch.equal(br.neg, bc.neg)
            for h in range(2):
                n, t, x = br.nbr_nids[h], br.nbr_edge_time[h], br.nbr_edge_x[h]
                assert torch.equal(n, bc.nbr_nids[h]) and torch.equal(t, bc.nbr_edge_time[h]) and torch.equal(x, bc.nbr_edge_x[h]), (n_seen, h)
                valid = n >= 0
                q = br.seed_times[h][:, None]
                assert bool(((t < q) | ~valid).all()) and bool((valid[:, 1:] | ~valid[:, :-1]).all())
                assert bool(((t[:, 1:] >= t[:, :-1]) | ~valid[:, :-1]).all())
                assert bool((t[~valid] == 0).all()) and bool((x[~valid] == 0).all())
                if h == 1:
                    n_valid += int(valid.sum())
            assert torch.equal(br.seed_nids[1], br.nbr_nids[0].reshape(-1))
            n_seen += 1
    hook_r.check()
    hook_c.check()
    assert n_seen == nb and n_valid > 0

"""NeighborSamplerHook on the device: exact where the reference is deterministic (<= k candidates), and for sampled rows
the properties the reference guarantees -- k distinct candidates from before the batch, the same row for every occurrence
of a node in a hop -- plus a uniformity check of the device generator."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DEV = torch.device('cuda', 0) if torch.cuda.is_available() else None
CASES = ['g11_uniform_sparse', 'g11_uniform_dense', 'g11_uniform_dense_directed']


def _load(case):
    z = np.load(os.path.join(HERE, 'golden', case + '.npz'))
    return json.loads(bytes(z['meta']).decode()), z


def _candidates(src, dst, ts, edge_x, node, ev_hi, directed):
    """[(nbr, time, feature bytes)] of `node` among the first ev_hi edges, in the reference's order."""
    out = []
    for i in range(ev_hi):
        fx = b'' if edge_x is None else edge_x[i].tobytes()
        if src[i] == node:
            out.append((int(dst[i]), int(ts[i]), fx))
        if not directed and dst[i] == node:
            out.append((int(src[i]), int(ts[i]), fx))
    return out


class _Replay:
    def __init__(self, neg):
        from tgm_amd.hooks.base import StatelessHook

        class R(StatelessHook):
            _cls_requires = {'edge_src', 'edge_dst', 'edge_time'}
            _cls_produces = {'neg', 'neg_time'}

            def __init__(s):
                super().__init__()
                s.__post_init__()

            def __call__(s, dg, batch):
                lo = batch._edge_lo
                batch.neg = neg[lo : lo + batch.edge_src.numel()].clone()
                batch.neg_time = batch.edge_time.clone()
                return batch

        self.hook = R()


@pytest.mark.parametrize('case', CASES)
def test_uniform_hook_against_reference(case):
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, NeighborSamplerHook

    meta, z = _load(case)
    src, dst, ts, neg = z['src'], z['dst'], z['ts'], z['neg']
    edge_x = z['edge_x'] if 'edge_x' in z.files else None
    ks, bs, directed = meta['num_nbrs'], meta['batch_size'], meta['directed']
    data = DGData.from_raw(torch.from_numpy(ts), torch.stack([torch.from_numpy(src), torch.from_numpy(dst)], 1),
                           None if edge_x is None else torch.from_numpy(edge_x))  # fmt: skip
    dg = DGraph(data, device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', _Replay(torch.from_numpy(neg).to(DEV)).hook)
    hm.register('k', NeighborSamplerHook(ks, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], directed=directed, seed=7))
    exact_rows = sampled_rows = 0
    with hm.activate('k'):
        for b, batch in enumerate(DGDataLoader(dg, batch_size=bs, hook_manager=hm)):
            lo = b * bs
            ev_hi = int(np.searchsorted(ts, ts[lo], side='left'))
            for h, k in enumerate(ks):
                g_seed = z[f'b{b}_h{h}_seed_nids']
                seeds = batch.seed_nids[h].cpu().numpy()
                n, t, x = batch.nbr_nids[h].cpu().numpy(), batch.nbr_edge_time[h].cpu().numpy(), batch.nbr_edge_x[h].cpu().numpy()
                assert n.shape == (len(seeds), k) and t.shape == n.shape and x.shape[:2] == n.shape
                if h == 0:
                    assert np.array_equal(seeds, g_seed)
                rows_of = {}
                for r, node in enumerate(seeds.tolist()):
                    cand = _candidates(src, dst, ts, edge_x, node, ev_hi, directed) if node >= 0 else []
                    got = [(int(n[r, j]), int(t[r, j]), x[r, j].tobytes() if x.shape[2] else b'') for j in range(k)]
                    pad = (-1, 0, np.zeros(x.shape[2], np.float32).tobytes() if x.shape[2] else b'')
                    if len(cand) <= k:
                        assert got == cand + [pad] * (k - len(cand)), f'b{b} h{h} row {r}'
                        if h == 0:  # hop-0 seeds are the reference's: the whole row must equal the golden one
                            assert np.array_equal(n[r], z[f'b{b}_h{h}_nbr_nids'][r]) and np.array_equal(t[r], z[f'b{b}_h{h}_nbr_edge_time'][r])
                        exact_rows += 1
                    else:
                        pool = list(cand)
                        for g in got:  # k distinct candidates (multiset inclusion), no padding
                            assert g in pool, f'b{b} h{h} row {r}: {g[:2]} is not a remaining candidate'
                            pool.remove(g)
                        sampled_rows += 1
                    if node in rows_of:  # every occurrence of a node in one hop gets the same row
                        assert got == rows_of[node]
                    rows_of[node] = got
    assert exact_rows > 0
    if 'dense' in case:
        assert sampled_rows > 0


def test_uniform_hook_deferred_validation_raises_at_check():
    """validate='deferred' (extension, like RecencyNeighborHook's): a seed outside [0, num_nodes) is still caught on the device, but the
    ValueError of the reference surfaces at ``hook.check()`` instead of costing every batch a device -> host read; 'sync' (default) raises in
    the call.  Valid batches give the same neighbours either way (same seed, same call counter)."""
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, NeighborSamplerHook, RandomNegativeEdgeSamplerHook

    g = torch.Generator().manual_seed(3)
    N, E = 50, 600
    ei = torch.randint(0, N, (E, 2), generator=g).int()
    data = DGData.from_raw(torch.arange(E), ei, torch.rand(E, 4, generator=g))

    def run(validate, bad):
        dg = DGraph(data, device=DEV)
        hm = HookManager(keys=['k'])
        hm.register('k', RandomNegativeEdgeSamplerHook(0, N + (40 if bad else 0), seed=5))  # bad: some negatives are not node ids
        hook = NeighborSamplerHook([4, 3], ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], seed=7, validate=validate)
        hm.register('k', hook)
        out = []
        with hm.activate('k'):
            for batch in DGDataLoader(dg, batch_size=100, hook_manager=hm):
                out.append([t.clone() for t in (batch.nbr_nids[0], batch.nbr_nids[1], batch.nbr_edge_time[1])])
        return hook, out

    hook_s, a = run('sync', False)
    hook_d, b = run('deferred', False)
    hook_d.check()
    assert all(torch.equal(x, y) for p, q in zip(a, b) for x, y in zip(p, q))
    with pytest.raises(ValueError, match='Seed nodes must satisfy'):
        run('sync', True)
    hook_d, _ = run('deferred', True)  # no error while iterating ...
    with pytest.raises(ValueError, match='Seed nodes must satisfy'):
        hook_d.check()  # ... it is raised here
    hook_d.check()  # (and cleared)
    with pytest.raises(ValueError, match='validate must be'):
        NeighborSamplerHook([2], ['edge_src'], ['edge_time'], validate='off')


def test_uniform_draw_is_uniform():
    """One hub with 12 candidates, k = 4, 3000 independent calls: every candidate is drawn ~1000 times, in every slot ~250."""
    from tgm_amd import _native
    from tgm_amd.index import build_csr

    E, N, k = 12, 20, 4
    src = torch.zeros(E, dtype=torch.int32, device=DEV)
    dst = torch.arange(1, E + 1, dtype=torch.int32, device=DEV)
    ts = torch.arange(1, E + 1, dtype=torch.int64, device=DEV)
    csr = build_csr(src, dst, ts, N, order='event', directed=True)
    lib = _native.load()
    calls = 3000
    seeds = torch.zeros(1, dtype=torch.int32, device=DEV)
    nid = torch.empty((calls, k), dtype=torch.int32, device=DEV)
    nts = torch.empty((calls, k), dtype=torch.int64, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    for c in range(calls):
        rc = lib.tgmx_uniform_lookup_csr(csr.indptr.data_ptr(), csr.adj.data_ptr(), None, 0, seeds.data_ptr(), 1, k, E, N, 0, 99, c << 8,
                                         nid[c].data_ptr(), nts[c].data_ptr(), None, status.data_ptr(), _native.stream_ptr(0))
        assert rc == 0
    got = nid.cpu().numpy()
    assert int(status.item()) == 0
    assert all(len(set(row)) == k for row in got.tolist())  # without replacement
    counts = np.bincount(got.reshape(-1), minlength=E + 1)[1:]
    sigma = np.sqrt(calls * (k / E) * (1 - k / E))
    assert np.all(np.abs(counts - calls * k / E) < 5 * sigma), counts
    for j in range(k):  # the order is random too
        cj = np.bincount(got[:, j], minlength=E + 1)[1:]
        sj = np.sqrt(calls / E * (1 - 1 / E))
        assert np.all(np.abs(cj - calls / E) < 5 * sj), (j, cj)
    # same (seed, stream, node) -> same row
    again = torch.empty((1, k), dtype=torch.int32, device=DEV)
    lib.tgmx_uniform_lookup_csr(csr.indptr.data_ptr(), csr.adj.data_ptr(), None, 0, seeds.data_ptr(), 1, k, E, N, 0, 99, 5 << 8,
                                again.data_ptr(), nts[0].data_ptr(), None, status.data_ptr(), _native.stream_ptr(0))
    assert np.array_equal(again.cpu().numpy()[0], got[5])


def test_uniform_draw_large_hub_k64():
    """A hub with 5000 candidates, k = 64 (the maximum), 300 calls: rows are 64 distinct candidates before ev_hi and the
    draws spread evenly over the candidate range."""
    from tgm_amd import _native
    from tgm_amd.index import build_csr

    E, N, k, ev_hi = 6000, 7000, 64, 5000
    src = torch.zeros(E, dtype=torch.int32, device=DEV)
    dst = torch.arange(1, E + 1, dtype=torch.int32, device=DEV)
    ts = torch.arange(1, E + 1, dtype=torch.int64, device=DEV)
    csr = build_csr(src, dst, ts, N, order='event', directed=True)
    lib = _native.load()
    calls = 300
    seeds = torch.zeros(1, dtype=torch.int32, device=DEV)
    nid = torch.empty((calls, k), dtype=torch.int32, device=DEV)
    nts = torch.empty((calls, k), dtype=torch.int64, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    for c in range(calls):
        assert lib.tgmx_uniform_lookup_csr(csr.indptr.data_ptr(), csr.adj.data_ptr(), None, 0, seeds.data_ptr(), 1, k, ev_hi, N, 0, 1234, c << 8,
                                           nid[c].data_ptr(), nts[c].data_ptr(), None, status.data_ptr(), _native.stream_ptr(0)) == 0
    got, t = nid.cpu().numpy(), nts.cpu().numpy()
    assert int(status.item()) == 0
    assert got.min() >= 1 and got.max() <= ev_hi  # only edges before ev_hi (dst = eid + 1)
    assert np.array_equal(t, got.astype(np.int64))  # times belong to the drawn edges
    assert all(len(set(row)) == k for row in got.tolist())
    counts = np.bincount((got.reshape(-1) - 1) * 10 // ev_hi, minlength=10)
    exp = calls * k / 10
    assert np.all(np.abs(counts - exp) < 5 * np.sqrt(exp * 0.9)), counts

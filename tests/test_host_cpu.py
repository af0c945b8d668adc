"""Host logic (no GPU): DGData validation, DGraph slicing, loader iteration, HookManager ordering,
hook constructor / seed errors -- modelled on the reference's own unit tests for these pieces."""
import numpy as np
import pytest
import torch

from tgm_amd import DGBatch, DGData, DGDataLoader, DGraph
from tgm_amd.core import TimeDeltaDG
from tgm_amd.exceptions import (BadHookProtocolError, EmptyGraphError, EventOrderedConversionError, InvalidNodeIDError,
                                UnresolvableHookDependenciesError)  # fmt: skip
from tgm_amd.hooks import DeduplicationHook, HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook, StatelessHook


def _data(E=50, N=9, D=3, seed=0, unit='r'):
    rng = np.random.default_rng(seed)
    ts = torch.from_numpy(np.sort(rng.integers(0, 40, E)).astype(np.int64))
    ei = torch.from_numpy(rng.integers(0, N, (E, 2)).astype(np.int32))
    return DGData.from_raw(ts, ei, torch.rand(E, D), time_delta=unit), ts, ei


def test_dgdata_normalises_and_validates():
    with pytest.warns(UserWarning):
        d = DGData.from_raw(torch.LongTensor([3, 1, 2]), torch.LongTensor([[0, 1], [1, 2], [2, 0]]), torch.rand(3, 2).double())
    assert d.edge_index.dtype == torch.int32 and d.edge_x.dtype == torch.float32 and d.time.dtype == torch.int64
    assert d.time.tolist() == [1, 2, 3] and d.edge_index.tolist() == [[1, 2], [2, 0], [0, 1]]  # re-sorted with payload
    with pytest.raises(InvalidNodeIDError):
        DGData.from_raw(torch.LongTensor([1]), torch.IntTensor([[0, -1]]))
    with pytest.raises(ValueError):
        DGData.from_raw(torch.LongTensor([-1]), torch.IntTensor([[0, 1]]))
    with pytest.raises(EmptyGraphError):
        DGData.from_raw(torch.LongTensor([]), torch.zeros((0, 2), dtype=torch.int32))
    with pytest.raises(ValueError):
        DGData.from_raw(torch.LongTensor([1, 2]), torch.IntTensor([[0, 1], [1, 2]]), torch.rand(3, 2))
    with pytest.raises(TypeError):
        DGraph('not data')


def test_slices_are_views_and_match_numpy():
    data, ts, ei = _data()
    dg = DGraph(data)
    assert dg.num_events == 50 and dg.num_edge_events == 50 and dg.edge_x_dim == 3
    assert dg.start_time == int(ts[0]) and dg.end_time == int(ts[-1])
    v = dg.slice_events(10, 25)
    b = v.materialize()
    assert b._edge_lo == 10 and torch.equal(b.edge_src, ei[10:25, 0]) and torch.equal(b.edge_time, ts[10:25])
    assert b.edge_src.data_ptr() == dg.edge_src[10:].data_ptr()  # zero-copy window
    t = dg.slice_time(5, 17)  # [5, 17)
    m = (ts >= 5) & (ts < 17)
    assert torch.equal(t.edge_time, ts[m]) and torch.equal(t.edge_x, data.edge_x[m])
    assert t.slice_events(0, 3).num_events <= 3
    with pytest.raises(ValueError):
        dg.slice_events(5, 2)
    assert dg.slice_time(1000, 2000).materialize().edge_src.numel() == 0


def test_loader_event_and_time_batches():
    data, ts, _ = _data(unit='s')
    dg = DGraph(data)
    sizes = [b.edge_src.numel() for b in DGDataLoader(dg, batch_size=8)]
    assert sum(sizes) == 50 and sizes[:-1] == [8] * 6 and len(DGDataLoader(dg, batch_size=8, drop_last=True)._starts) == 6
    got = torch.cat([b.edge_time for b in DGDataLoader(dg, batch_size=5, batch_unit='s')])
    assert torch.equal(got, ts)  # time windows of 5 s cover everything exactly once, empty ones skipped
    with pytest.raises(ValueError):
        DGDataLoader(dg, batch_size=0)
    with pytest.raises(EventOrderedConversionError):
        DGDataLoader(DGraph(_data()[0]), batch_size=2, batch_unit='s')
    assert TimeDeltaDG('h').convert('m') == 60 and TimeDeltaDG('m', 30).is_coarser_than('s')


class _Produces(StatelessHook):
    def __init__(self, req, prod, log, name):
        super().__init__()
        self._cls = name
        self._requires |= set(req)
        self._produces |= set(prod)
        self.log = log

    def __call__(self, dg, batch):
        self.log.append(self._cls)
        for p in self.produces:
            setattr(batch, p, torch.zeros(1))
        return batch


def test_hook_manager_orders_by_dependencies_and_neg_before_nbr():
    data, _, _ = _data()
    dg = DGraph(data)
    log = []
    hm = HookManager(keys=['a', 'b'])
    nbr = _Produces({'edge_src'}, {'nbr_nids'}, log, 'nbr')
    neg = _Produces({'edge_dst'}, {'neg'}, log, 'neg')
    user = _Produces({'nbr_nids', 'neg'}, {'z'}, log, 'user')
    hm.register('a', user)
    hm.register('a', nbr)
    hm.register('a', neg)  # registered last, must still run before the neighbor hook
    with pytest.raises(RuntimeError):
        hm.execute_active_hooks(dg, dg.materialize())
    with hm.activate('a'):
        hm.execute_active_hooks(dg, dg.materialize())
        with pytest.raises(RuntimeError):
            hm.register('a', neg)
    assert log == ['neg', 'nbr', 'user']
    with pytest.raises(KeyError):
        hm.set_active_hooks('nope')
    with pytest.raises(BadHookProtocolError):
        hm.register('a', object())
    hm2 = HookManager(keys=['k'])
    hm2.register('k', _Produces({'never_made'}, {'q'}, log, 'x'))
    with pytest.raises(UnresolvableHookDependenciesError):
        hm2.resolve_hooks()
    hm3 = HookManager(keys=['k'])  # cycle
    hm3.register('k', _Produces({'p'}, {'q'}, log, 'x'))
    hm3.register('k', _Produces({'q'}, {'p'}, log, 'y'))
    with pytest.raises(UnresolvableHookDependenciesError):
        hm3.resolve_hooks('k')
    with pytest.raises(ValueError):
        HookManager(keys=[])


def test_recency_hook_contract_on_host():
    h = RecencyNeighborHook(num_nbrs=[1], num_nodes=1, seed_nodes_keys=['edge_src'], seed_times_keys=['edge_time'])
    assert h.has_state and h.requires == {'edge_src', 'edge_dst', 'edge_time'}
    assert h.produces == {'seed_nids', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x', 'seed_times', 'seed_node_nbr_mask'}
    hid = RecencyNeighborHook(num_nbrs=[1], num_nodes=1, seed_nodes_keys=['foo'], seed_times_keys=['bar'], id='x')
    assert hid.produces == {p + '_x' for p in h.produces} and 'foo' in hid.requires and 'x' in repr(hid)
    for bad in ([0], [-1], []):
        with pytest.raises(ValueError):
            RecencyNeighborHook(num_nbrs=bad, num_nodes=2, seed_nodes_keys=['a'], seed_times_keys=['b'])
    with pytest.raises(ValueError):
        RecencyNeighborHook(num_nbrs=[1], num_nodes=2, seed_nodes_keys=['a', 'b'], seed_times_keys=['a'])
    with pytest.raises(ValueError):
        RecencyNeighborHook(num_nbrs=[1], num_nodes=2, seed_nodes_keys=['a'], seed_times_keys=['a'], mode='csr')
    # host-side seed checks (structure + values) run before any device work
    data, _, _ = _data()
    dg = DGraph(data)
    hook = RecencyNeighborHook(num_nbrs=[1], num_nodes=2, seed_nodes_keys=['foo'], seed_times_keys=['bar'])
    batch = dg.materialize()
    with pytest.raises(ValueError):
        hook(dg, batch)  # missing attributes
    batch.foo, batch.bar = torch.IntTensor([5]), torch.LongTensor([1])
    with pytest.raises(ValueError):
        hook(dg, batch)  # id out of range
    batch.foo, batch.bar = torch.IntTensor([1]), torch.LongTensor([-1])
    with pytest.raises(ValueError):
        hook(dg, batch)  # negative time
    batch.foo, batch.bar = None, None
    with pytest.warns(UserWarning):
        out = hook(dg, batch)  # None seeds: empties, no device needed
    assert out.nbr_nids[0].numel() == 0 and out.nbr_edge_x[0].shape == (0, 3)


def test_negative_and_dedup_hooks_on_host():
    data, _, ei = _data()
    dg = DGraph(data)
    b = RandomNegativeEdgeSamplerHook(3, 9, seed=1)(dg, dg.slice_events(0, 7).materialize())
    assert b.neg.dtype == torch.int32 and b.neg.shape == (7,) and int(b.neg.min()) >= 3 and int(b.neg.max()) < 9
    assert torch.equal(b.neg_time, b.edge_time) and b.neg_time.data_ptr() != b.edge_time.data_ptr()
    with pytest.raises(ValueError):
        RandomNegativeEdgeSamplerHook(5, 5)
    b.nbr_nids = [torch.IntTensor([[8, -1], [2, 2]])]
    b = DeduplicationHook(['neg', 'nbr_nids'])(dg, b)
    exp = torch.unique(torch.cat([b.edge_src, b.edge_dst, b.neg, torch.IntTensor([8, 2, 2])]))
    assert torch.equal(b.unique_nids, exp) and torch.equal(b.unique_nids[b.global_to_local(b.edge_src).long()], b.edge_src)
    assert 'edge_src = [7]' in str(b) and isinstance(b, DGBatch)


def test_discretize_matches_reference_golden():
    """DGData.discretize against the reference's output (golden g9): bit-exact, all event groups."""
    import os

    from tgm_amd.exceptions import InvalidDiscretizationError

    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g9_discretize.npz'))
    T = torch.from_numpy
    d = DGData.from_raw(T(z['ts']), T(z['ei']), T(z['ex']), node_x_time=T(z['nt']), node_x_nids=T(z['nn']), node_x=T(z['nx']),
                        node_y_time=T(z['yt']), node_y_nids=T(z['yn']), node_y=T(z['yv']), time_delta='s')  # fmt: skip
    def canon(times, *cols):
        """rows sorted lexicographically inside each timestamp bucket: the reference orders events that
        share a (discretized) timestamp with a NON-stable argsort (dg_data.py:360), i.e. unspecified;
        ours keeps the input order.  Everything else -- which events survive, their buckets, payloads -- is exact."""
        rows = np.concatenate([np.asarray(times, np.float64)[:, None]] + [np.asarray(c, np.float64).reshape(len(times), -1) for c in cols], 1)
        return rows[np.lexsort(rows.T[::-1])]

    for unit in ('m', 'h'):
        with pytest.warns(UserWarning):  # bucketed node events precede later edges: the timeline is re-sorted
            c = d.discretize(unit)
        assert c.time_delta == TimeDeltaDG(unit)
        assert np.array_equal(c.time.numpy(), z[f'{unit}_time']) and c.time.dtype == torch.int64
        for mask, cols in (('edge_mask', ('edge_index', 'edge_x')), ('node_x_mask', ('node_x_nids', 'node_x')), ('node_y_mask', ('node_y_nids', 'node_y'))):
            t_got = c.time[getattr(c, mask).long()].numpy()
            t_exp = z[f'{unit}_time'][z[f'{unit}_{mask}']]
            got = canon(t_got, *[getattr(c, f).numpy() for f in cols])
            exp = canon(t_exp, *[z[f'{unit}_{f}'] for f in cols])
            assert got.shape == exp.shape and np.array_equal(got, exp), f'{unit} {mask}'
            for f in cols:
                assert getattr(c, f).numpy().dtype == z[f'{unit}_{f}'].dtype
        assert np.array_equal(np.sort(np.concatenate([c.edge_mask.numpy(), c.node_x_mask.numpy(), c.node_y_mask.numpy()])), np.arange(len(c.time)))
    same = d.discretize('s')
    assert same is not d and torch.equal(same.time, d.time)
    with pytest.raises(InvalidDiscretizationError):
        d.discretize('h').discretize('m')
    with pytest.raises(EventOrderedConversionError):
        _data()[0].discretize('h')
    with pytest.raises(ValueError):
        d.discretize('h', reduce_op='mean')
    # a coarser graph iterates one snapshot per bucket
    dg = DGraph(d.discretize('h'))
    sizes = [b.edge_src.numel() for b in DGDataLoader(dg, batch_unit='h')]
    assert sum(sizes) == dg.num_edge_events and len(sizes) >= 5


def test_tgmx_slice_matches_the_store_s_event_range():
    """tgmx_slice (host arithmetic in the C ABI) == EdgeStore.event_range == the reference's _binary_search
    (array_backend.py:301-321) on a tie-heavy timeline, every bound combination."""
    import ctypes

    import numpy as np

    from tgm_amd import _native
    from tgm_amd.core.store import SliceBounds

    lib = _native.load()
    rng = np.random.default_rng(3)
    t = np.sort(rng.integers(0, 40, 300)).astype(np.int64)
    lb, ub = ctypes.c_int64(), ctypes.c_int64()

    class FakeStore:
        _time_np = t
        num_events = len(t)

    from tgm_amd.core.store import EdgeStore

    for _ in range(400):
        st_ = None if rng.random() < 0.3 else int(rng.integers(-3, 45))
        et_ = None if rng.random() < 0.3 else int(rng.integers(-3, 45))
        si_ = None if rng.random() < 0.5 else int(rng.integers(0, 300))
        ei_ = None if rng.random() < 0.5 else int(rng.integers(1, 301))
        want = EdgeStore.event_range(FakeStore, SliceBounds(st_, et_, si_, ei_))
        rc = lib.tgmx_slice(t.ctypes.data, len(t), st_ is not None, st_ or 0, et_ is not None, et_ or 0, si_ or 0, ei_ if ei_ else -1,
                            ctypes.byref(lb), ctypes.byref(ub))
        assert rc == 0 and (lb.value, ub.value) == want, (st_, et_, si_, ei_)


def test_reference_side_binding_script():
    """INTEGRATION.md section 2 as an executed check: our hooks inside the REFERENCE's HookManager (protocol, registration,
    neg -> nbr ordering edge), identical requires / produces, DGBatch fields, TGAT state_dict interchange.  Needs the
    reference checkout, so it only runs in the build container (skipped elsewhere; nothing on the GPU box reads it)."""
    import os
    import subprocess
    import sys

    if not os.path.isdir('/root/reference/tgm'):
        pytest.skip('reference checkout not present')
    script = os.path.join(os.path.dirname(__file__), 'golden', 'check_reference_binding.py')
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count('[binding] ok') == 6
    import re

    m = re.search(r'reference unit tests against tgm_amd: (\d+) / (\d+)', r.stdout)
    assert m and m.group(1) == m.group(2) and int(m.group(2)) >= 210, r.stdout


def test_loader_is_a_torch_dataloader_like_the_reference():
    """tgm/data/loader.py:64,147-149: the reference subclasses torch.utils.data.DataLoader and forwards **kwargs to it.  Ours
    does too (isinstance, len, dataset = the slice starts, torch validates the keyword arguments), and refuses the one setting
    that cannot work for either implementation: worker processes (hook state would fork)."""
    import torch.utils.data

    from tgm_amd import DGData, DGDataLoader, DGraph

    ts = torch.arange(50)
    src, dst = torch.randint(0, 10, (50,), dtype=torch.int32), torch.randint(0, 10, (50,), dtype=torch.int32)
    dg = DGraph(DGData.from_raw(ts, torch.stack([src, dst], 1), torch.rand(50, 3)))
    loader = DGDataLoader(dg, batch_size=7, drop_last=True, pin_memory=False, timeout=0)
    assert isinstance(loader, torch.utils.data.DataLoader)
    assert len(loader) == 7 and list(loader.dataset) == list(range(0, 43, 7)) and loader.drop_last and loader.collate_fn is loader
    assert sum(b.edge_src.numel() for b in loader) == 49
    assert len(DGDataLoader(dg, batch_size=7)) == 8
    with pytest.raises(TypeError, match='bogus'):
        DGDataLoader(dg, batch_size=7, bogus=1)
    with pytest.raises(ValueError, match='num_workers'):
        DGDataLoader(dg, batch_size=7, num_workers=2)


def test_parameter_cache_key_sees_fused_optimizer_steps_and_explicit_invalidation():
    """Derived weight copies are cached against tgm_amd.nn._paramver.param_key.  A fused optimizer step leaves Tensor._version alone
    (which is why the key carries an optimizer-step count); writes through .data need invalidate_parameter_caches()."""
    import torch

    from tgm_amd.nn import invalidate_parameter_caches
    from tgm_amd.nn._paramver import param_key

    lin = torch.nn.Linear(4, 4)
    for make in (lambda: torch.optim.Adam(lin.parameters(), lr=1e-3, fused=True), lambda: torch.optim.Adam(lin.parameters(), lr=1e-3),
                 lambda: torch.optim.SGD(lin.parameters(), lr=1e-3, fused=True)):  # fmt: skip
        opt = make()
        for p in lin.parameters():
            p.grad = torch.ones_like(p)
        k0 = param_key(lin.parameters())
        assert param_key(lin.parameters()) == k0  # stable between steps
        opt.step()
        assert param_key(lin.parameters()) != k0
    k0 = param_key(lin.parameters())
    lin.weight.data.mul_(0.5)
    assert param_key(lin.parameters()) == k0  # the blind spot ...
    invalidate_parameter_caches()
    assert param_key(lin.parameters()) != k0  # ... and its remedy
    k0 = param_key(lin.parameters())
    lin.load_state_dict({k: v + 1 for k, v in lin.state_dict().items()})
    assert param_key(lin.parameters()) != k0


def test_modules_deep_copy_and_pickle_without_their_derived_caches():
    """The modules cache ctypes blocks (pointers into device buffers) and workspaces on themselves; ctypes structures with pointers cannot
    be pickled, so ``copy.deepcopy(model)`` / ``torch.save(model)`` after a forward used to raise.  Derived state stays behind."""
    import copy
    import io

    import torch

    from tgm_amd import _native
    from tgm_amd.nn import TGAT, GraphAttentionEmbedding, IdentityMessage, LastAggregator, TGNMemory, Time2Vec

    enc = TGAT(node_dim=3, edge_dim=4, time_dim=6, embed_dim=8, num_layers=2)
    enc._desc_cache = ('key', (_native.TgatModel(), [torch.zeros(3)]))  # what a forward leaves behind
    enc._desc_struct = ('skey', _native.TgatModel(), [torch.zeros(3)], [_native.PackJob()])
    enc._workspace = torch.zeros(16)
    enc.__dict__['_tgmx_plist'] = (0, tuple(enc.parameters()))
    mem = TGNMemory(10, 4, 8, 6, IdentityMessage(4, 8, 6), LastAggregator())
    mem._fwd_args = _native.TgnMemoryFwd()
    gae = GraphAttentionEmbedding(8, 8, 4, Time2Vec(6))
    gae.conv._fwd_args = _native.TconvFwd()
    for mod in (enc, mem, gae):
        twin = copy.deepcopy(mod)
        buf = io.BytesIO()
        torch.save(mod, buf)
        buf.seek(0)
        again = torch.load(buf, weights_only=False)
        for other in (twin, again):
            assert all(torch.equal(a, b) and a is not b for a, b in zip(mod.state_dict().values(), other.state_dict().values()))
    twin = copy.deepcopy(enc)
    assert twin._desc_cache is None and twin._desc_struct is None and twin._workspace is None and '_tgmx_plist' not in twin.__dict__
    assert enc._desc_cache is not None  # the original keeps its own


def test_rings_under_a_process_group_warn_once_and_name_the_static_index(monkeypatch):
    """RecencyNeighborHook(mode='ring') under torch.distributed with more than one rank: one UserWarning that names mode='csr' and
    batch_shard (the multi-GPU mode); nothing for a single rank."""
    import warnings

    import torch.distributed as dist

    from tgm_amd.hooks import RecencyNeighborHook

    hook = RecencyNeighborHook(10, [2], ['edge_src'], ['edge_time'])
    monkeypatch.setattr(RecencyNeighborHook, '_warned_world', False, raising=False)
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_world_size', lambda *a, **k: 1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        hook._warn_rings_under_world()
    assert not w
    monkeypatch.setattr(dist, 'get_world_size', lambda *a, **k: 8)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        hook._warn_rings_under_world()
        hook._warn_rings_under_world()
    assert len(w) == 1 and "mode='csr'" in str(w[0].message) and 'batch_shard' in str(w[0].message)


def test_sampler_call_tag_survives_inference_tensors():
    """``Tensor._version`` raises on inference tensors; a sampler call made under ``torch.inference_mode()`` must still publish (its tag
    then matches nothing, i.e. the consumer takes the row-per-slot computation)."""
    import torch

    from tgm_amd.core.lazy import SampledHops, SamplerCallTag

    with torch.inference_mode():
        ids, times = [torch.zeros(4, 2, dtype=torch.int32)], [torch.zeros(4, 2, dtype=torch.int64)]
        tag = SamplerCallTag(ids, times)
        assert not tag.matches(ids, times)
        assert SampledHops(ids, tag).copy().tag is tag
    ids, times = [torch.zeros(4, 2, dtype=torch.int32)], [torch.zeros(4, 2, dtype=torch.int64)]
    tag = SamplerCallTag(ids, times)
    assert tag.matches(ids, times)
    ids[0].add_(1)
    assert not tag.matches(ids, times)


def test_splits_are_contiguous_ranges_equal_to_the_mask_formulation():
    """``DGData.split`` (tgm/data/dg_data.py:396-421, tgm/data/split.py:99-243): the contiguous-range cut of the sorted timeline against the
    reference's formulation (a boolean mask per event group and interval) on a stream with node events, node labels and timestamp ties; a
    split without edges is dropped, one without node events carries None; ratio splits cut the time span; a shipped split cannot be replaced."""
    from tgm_amd.data import TemporalRatioSplit, TemporalSplit, TGBSplit

    g = torch.Generator().manual_seed(5)
    E, NX, NY, N = 300, 40, 30, 25
    ets = torch.sort(torch.randint(10, 200, (E,), generator=g)).values
    ei = torch.randint(0, N, (E, 2), generator=g).int()
    ex = torch.rand(E, 3, generator=g)
    xts, xid, xv = torch.randint(0, 150, (NX,), generator=g), torch.randint(0, N, (NX,), generator=g).int(), torch.rand(NX, 2, generator=g)
    yts, yid, yv = torch.randint(0, 220, (NY,), generator=g), torch.randint(0, N, (NY,), generator=g).int(), torch.rand(NY, 4, generator=g)
    ei[0] = torch.tensor([N - 1, N - 1], dtype=torch.int32)  # (labels may only name ids the edges / node events span)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        data = DGData.from_raw(ets, ei, ex, xts, xid, xv, yts, yid, yv, static_node_x=torch.rand(N, 2, generator=g), time_delta='s')

    def expect(lo, hi):
        t_e, t_x, t_y = data.time[data.edge_mask.long()], data.time[data.node_x_mask.long()], data.time[data.node_y_mask.long()]
        m_e, m_x, m_y = (t_e >= lo) & (t_e < hi), (t_x >= lo) & (t_x < hi), (t_y >= lo) & (t_y < hi)
        return (data.edge_index[m_e], data.edge_x[m_e], t_e[m_e], data.node_x_nids[m_x], data.node_x[m_x], t_x[m_x], data.node_y_nids[m_y], data.node_y[m_y], t_y[m_y])

    def check(part, lo, hi):
        # (the part goes through DGData.from_raw like the reference's: its merged timeline -- edges | node events | labels -- is re-sorted there)
        e_i, e_x, e_t, x_i, x_v, x_t, y_i, y_v, y_t = expect(lo, hi)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ref = DGData.from_raw(e_t, e_i, e_x, x_t if x_i.numel() else None, x_i if x_i.numel() else None, x_v if x_i.numel() else None,
                                  y_t if y_i.numel() else None, y_i if y_i.numel() else None, y_v if y_i.numel() else None,
                                  static_node_x=data.static_node_x, time_delta='s')
        for f in ('time', 'edge_mask', 'edge_index', 'edge_x', 'node_x_mask', 'node_x_nids', 'node_x', 'node_y_mask', 'node_y_nids', 'node_y'):
            u, v = getattr(part, f), getattr(ref, f)
            assert (u is None and v is None) or torch.equal(u, v), f
        assert part.static_node_x is data.static_node_x and part.time_delta == data.time_delta
        assert sorted(part.time[part.edge_mask.long()].tolist()) == sorted(e_t.tolist())

    train, val, test = data.split(TemporalSplit(val_time=120, test_time=170))
    check(train, -1, 120), check(val, 120, 170), check(test, 170, 10**9)
    assert test.node_x_nids is None  # node events end at t < 150
    first, last = int(data.time[0]), int(data.time[-1])
    span = last - first + 1
    vt = first + int(span * 0.7)
    parts = data.split()  # the default: 70 / 15 / 15 % of the time span
    assert len(parts) == 3
    check(parts[0], -1, vt), check(parts[1], vt, vt + int(span * 0.15)), check(parts[2], vt + int(span * 0.15), 10**9)
    assert sum(p.edge_index.shape[0] for p in parts) == E
    assert len(data.split(TemporalSplit(val_time=500, test_time=600))) == 1  # no edge at or after t = 500: two splits do not exist
    with pytest.raises(ValueError):
        TemporalSplit(val_time=5, test_time=4)
    with pytest.raises(ValueError):
        TemporalRatioSplit(0.5, 0.2, 0.2)
    shipped = TGBSplit({'train': (10, 99), 'val': (100, 149), 'test': (150, 199)})
    data._split_strategy = shipped
    a, b, c = data.split()
    assert a.edge_index.shape[0] + b.edge_index.shape[0] + c.edge_index.shape[0] == E and int(b.time[b.edge_mask.long()].min()) >= 100
    with pytest.raises(ValueError, match='Cannot override'):
        data.split(TemporalRatioSplit())

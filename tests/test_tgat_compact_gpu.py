"""Compact rows (ABI v5, ``tgmx_tgat_hop_t.seed_keyed``): inference over the DISTINCT (id, time) rows of every level of the hop tree.

Hop h + 1 is seeded with hop h's flattened outputs and a seed's window is a function of (id, time) (tgm/hooks/neighbors/
recency.py:141-143, 161-163), so slots with equal (id, time) are the same row of every deeper level; tgm/nn/encoder/tgat.py:128-136
computes them once per slot all the same.  The compact forward must give the row-per-slot forward's embeddings BIT FOR BIT
(a row's arithmetic does not depend on where it sits), and must not engage for inputs that carry no sampler tag."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('n,n_ids,pad_frac,t_hi', [(1, 5, 0.0, 10), (63, 4, 0.5, 3), (5000, 40, 0.4, 50), (12000, 9000, 0.63, 2_600_000), (200_000, 3000, 0.3, 1 << 40)])
def test_pair_dedup_matches_numpy_unique(n, n_ids, pad_frac, t_hi):
    from tgm_amd import _native

    lib = _native.load()
    g = torch.Generator().manual_seed(n)
    ids = torch.randint(0, n_ids, (n,), generator=g, dtype=torch.int32)
    ts = torch.randint(0, t_hi, (n,), generator=g, dtype=torch.int64)
    pad = torch.rand(n, generator=g) < pad_frac
    ids[pad], ts[pad] = -1, 0
    if n > 100:  # pads with another time are pairs of their own
        ids[7], ts[7] = -1, 5
    d_ids, d_ts = ids.to(DEV), ts.to(DEV)
    uniq, owner, cidx = (torch.full((n,), -7, dtype=torch.int32, device=DEV) for _ in range(3))
    count = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    ws = torch.empty(int(lib.tgmx_pair_dedup_workspace_bytes(n)), dtype=torch.uint8, device=DEV)
    ws.random_(0, 255)  # contents irrelevant on entry
    for _ in range(2):  # twice into the same buffers: the second call must not see the first one's table
        _native.check(lib.tgmx_pair_dedup(d_ids.data_ptr(), d_ts.data_ptr(), n, uniq.data_ptr(), owner.data_ptr(), cidx.data_ptr(), count.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _native.stream_ptr()), 'tgmx_pair_dedup')
    torch.cuda.synchronize()
    key = np.stack([ids.numpy().astype(np.int64), ts.numpy()], 1)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    c = int(count.item())
    assert c == len(first)
    own = owner.cpu().numpy()
    assert np.array_equal(own, first[inv.reshape(-1)]), 'owner = the smallest index carrying the pair'
    u = uniq.cpu().numpy()[:c]
    assert sorted(u.tolist()) == sorted(first.tolist())  # a dense numbering of the owners, order unspecified
    ci = cidx.cpu().numpy()
    assert np.array_equal(u[ci[u]], u) and np.array_equal(ci[u], np.arange(c))


def _pipeline(st, k, mode, features, bs=200, pool=None):
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook

    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x, static_node_x=st.node_x), device=DEV)
    hm = HookManager(keys=['k'])
    hm.register('k', RandomNegativeEdgeSamplerHook(int(st.dst.min()), st.num_nodes, seed=4))
    hm.register('k', RecencyNeighborHook(st.num_nodes, k, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], mode=mode,
                                         batch_size=bs if mode == 'csr' else None, edge_features=features))
    return dg, hm, DGDataLoader(dg, batch_size=bs, hook_manager=hm, output_pool=pool)


def _plain(batch):
    """the same tensors without the sampler's tag: what a hand-made input looks like -> row-per-slot computation"""
    x = batch.nbr_edge_x
    return (list(batch.seed_nids), list(batch.seed_times), list(batch.nbr_nids), x if hasattr(x, 'eids') else list(x), list(batch.nbr_edge_time))


def _engaged(enc, dg, batch):
    """did the last forward take the compact path?  (the layout of the same call says so)"""
    from tgm_amd import _native

    lay = _native.TgatLayout()
    hops = enc._last_hops
    _native.check(_native.load().tgmx_tgat_layout(enc._desc_cache[1][0], batch.seed_nids[0].numel(), hops, 0, lay), 'layout')
    return lay.compact[1] >= 0


@pytest.mark.parametrize('mode', ['ring', 'csr'])
@pytest.mark.parametrize('features', ['dense', 'by_id'])
def test_compact_rows_equal_row_per_slot_at_the_headline_dims(mode, features):
    from tgm_amd.nn import TGAT
    from tgm_amd.synth import make_stream

    st = make_stream('wiki', seed=3, num_edges=16_000, edge_dim=172)
    dg, hm, loader = _pipeline(st, [20, 20], mode, features)
    torch.manual_seed(0)
    enc = TGAT(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2).to(DEV).eval()
    enc.compact_rows = True
    with torch.no_grad():
        for p in enc.parameters():
            p.add_(0.03 * torch.randn_like(p))
    node_x = dg.static_node_x
    checked = 0
    with hm.activate('k'), torch.no_grad():
        for n, b in enumerate(loader):
            if n % 13 not in (0, 5) and n != 79:
                continue
            z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
            assert enc._last_hops[1].seed_keyed == 1 and _engaged(enc, dg, b)
            sn, stt, nn_, nx, nt = _plain(b)
            if hasattr(nx, 'eids'):
                nx.tag = None
            z_ref = enc(node_x, sn, stt, nn_, nx, nt)
            assert enc._last_hops[1].seed_keyed == 0
            assert torch.equal(z, z_ref), f'batch {n}: max |d| = {(z - z_ref).abs().max().item():.3e}'
            checked += 1
    assert checked >= 12


def test_compact_rows_three_layers_and_narrow_k():
    """L = 3 (two deduplicated levels in layer 1's row batch, one in layer 2's), k = 5, other widths (node_dim 8: float4 neighbor rows)."""
    from tgm_amd.nn import TGAT
    from tgm_amd.synth import make_stream

    st = make_stream('wiki', seed=9, num_edges=9_000, edge_dim=12, node_dim=8, n_src=300, n_dst=40)
    dg, hm, loader = _pipeline(st, [5, 5, 5], 'ring', 'dense', bs=150)
    torch.manual_seed(1)
    enc = TGAT(node_dim=8, edge_dim=12, time_dim=16, embed_dim=32, num_layers=3).to(DEV).eval()
    enc.compact_rows = True
    node_x = dg.static_node_x
    with hm.activate('k'), torch.no_grad():
        for n, b in enumerate(loader):
            if n % 7:
                continue
            z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
            assert _engaged(enc, dg, b)
            z_ref = enc(node_x, *_plain(b))
            assert torch.equal(z, z_ref), f'batch {n}'
    assert n >= 50


def test_untagged_or_modified_inputs_get_the_row_per_slot_computation():
    """The promise is the sampler's, not the tensors': plain lists, a replaced item, a tensor modified in place through torch and
    hop lists that are not seeded by one another all run row per slot -- where rows with equal (id, time) MAY differ."""
    from tgm_amd.nn import TGAT
    from tgm_amd.synth import make_stream

    st = make_stream('wiki', seed=5, num_edges=4_000, edge_dim=8, n_src=200, n_dst=30)
    dg, hm, loader = _pipeline(st, [4, 4], 'ring', 'dense', bs=100, pool=0)
    torch.manual_seed(2)
    enc = TGAT(node_dim=1, edge_dim=8, time_dim=8, embed_dim=16, num_layers=2).to(DEV).eval()
    enc.compact_rows = True
    node_x = dg.static_node_x
    with hm.activate('k'), torch.no_grad():
        for n, b in enumerate(loader):
            if n == 30:
                break
        args = lambda: (node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
        z = enc(*args())
        assert enc._last_hops[1].seed_keyed == 1
        # a hop-1 window that is NOT a function of its seed: perturb one of two rows that share a seed
        ids0 = b.nbr_nids[0].view(-1)
        t0 = b.nbr_edge_time[0].view(-1)
        key = ids0.long() * (1 << 40) + t0
        vals, counts = torch.unique(key[ids0 >= 0], return_counts=True)
        assert (counts > 1).any(), 'the test stream should repeat a (neighbor, time) pair'
        dup = (key == vals[counts > 1][0]).nonzero().view(-1)
        b.nbr_edge_x[1][dup[1]] += 1.0  # in place: bumps the version -> the tag no longer matches
        z2 = enc(*args())
        assert enc._last_hops[1].seed_keyed == 0
        ref = enc(node_x, *_plain(b))
        assert torch.equal(z2, ref)
        seed_row = int(dup[1]) // 4
        assert not torch.equal(z2[seed_row], z[seed_row]), 'the perturbed row must show in its seed (row per slot)'
    # a replaced list item, and plain lists
    with hm.activate('k'), torch.no_grad():
        b2 = next(iter(loader))
        z_t = enc(node_x, b2.seed_nids, b2.seed_times, b2.nbr_nids, b2.nbr_edge_x, b2.nbr_edge_time)
        assert enc._last_hops[1].seed_keyed == 1
        b2.nbr_nids[1] = b2.nbr_nids[1].clone()
        z_r = enc(node_x, b2.seed_nids, b2.seed_times, b2.nbr_nids, b2.nbr_edge_x, b2.nbr_edge_time)
        assert enc._last_hops[1].seed_keyed == 0 and torch.equal(z_t, z_r)
        enc(node_x, *_plain(b2))
        assert enc._last_hops[1].seed_keyed == 0


@pytest.mark.parametrize('features', ['dense', 'by_id'])
@pytest.mark.parametrize('compact', [False, True])
def test_in_kernel_folded_queries_equal_the_qf_buffer(features, compact):
    """ABI v6, ``tgmx_tgat_layer_t.qf_lane``: a layer whose rows carry ONE input feature has its folded queries evaluated inside the
    attention kernel (no [rows, H * p4(C)] buffer).  Same fma per column as the kernel that fills the buffer: the
    embeddings must be equal bit for bit with the table withheld."""
    from tgm_amd.nn import TGAT
    from tgm_amd.synth import make_stream

    st = make_stream('wiki', seed=5, num_edges=8_000, edge_dim=172)
    dg, hm, loader = _pipeline(st, [20, 20], 'ring', features)
    torch.manual_seed(2)
    enc = TGAT(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2).to(DEV).eval()
    enc.compact_rows = compact
    node_x = dg.static_node_x
    checked = 0
    with hm.activate('k'), torch.no_grad():
        for n, b in enumerate(loader):
            if n % 9:
                continue
            args = (node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
            z = enc(*args)
            model = enc._desc_cache[1][0]
            table = model.layers[0].qf_lane
            assert table, 'node_dim = 1: the lane-ordered table exists'
            assert not model.layers[1].qf_lane, 'layer 2 reads 172 features per row: a qf buffer'
            model.layers[0].qf_lane = None
            try:
                z_buf = enc(*args)
            finally:
                model.layers[0].qf_lane = table
            assert torch.equal(z, z_buf), f'batch {n}: max |d| = {(z - z_buf).abs().max().item():.3e}'
            checked += 1
    assert checked >= 4


def test_both_heads_in_one_pass_equal_the_per_head_passes():
    """ABI v6, ``tgmx_tgat_layer_t.W_V_t16c``: the one-kernel tail's first stage over both heads' output blocks at once (the stacked,
    per-head padded W_V as one tiled matrix).  Every output block still accumulates its k-steps in ascending order: the embeddings must
    equal the per-head passes' bit for bit."""
    from tgm_amd.nn import TGAT
    from tgm_amd.synth import make_stream

    st = make_stream('wiki', seed=8, num_edges=8_000, edge_dim=172)
    dg, hm, loader = _pipeline(st, [20, 20], 'ring', 'by_id')
    torch.manual_seed(4)
    enc = TGAT(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2).to(DEV).eval()
    with torch.no_grad():
        for p in enc.parameters():
            p.add_(0.03 * torch.randn_like(p))
    node_x = dg.static_node_x
    checked = 0
    with hm.activate('k'), torch.no_grad():
        for n, b in enumerate(loader):
            if n % 9 != 4:
                continue
            args = (node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
            z = enc(*args)
            model = enc._desc_cache[1][0]
            image = model.layers[0].W_V_t16c
            assert image, 'two heads of 51 columns: 2 x 4 blocks fit the 12 a stage may have'
            model.layers[0].W_V_t16c = None
            try:
                z_per_head = enc(*args)
            finally:
                model.layers[0].W_V_t16c = image
            assert torch.equal(z, z_per_head), f'batch {n}: max |d| = {(z - z_per_head).abs().max().item():.3e}'
            checked += 1
    assert checked >= 4

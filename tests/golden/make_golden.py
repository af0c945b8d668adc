#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference; PyG is replaced by the
names-only placeholder in tests/golden/_pyg_stub).  Nothing here travels as
code: the outputs are plain .npz data -- inputs + the reference's outputs -- and
the tests compare the oracle and the HIP path against them.

    python tests/golden/make_golden.py            # (re)write every fixture

Fixture families:
  g1_*   the reference's own hand-built sampler graphs (its unit-test inputs)
  g2_*   random small streams: heavy timestamp ties, self loops, non-bipartite,
         directed / undirected, unequal k per hop, negatives as a third seed group
  g3_*   wiki-shaped medium stream, k=[20,20], bs=200: SHA-256 of every output
  g4_*   epoch boundary (reset_state) and train -> val carry-over
  g5_*   TemporalAttention / TGAT eval-mode forward
  g5_self_noise.json  the reference against itself on the g5 fixtures (thread count, float64): the scale of the float bar
  g6_*   Time2Vec on int64 deltas up to 2^31
  g7_*   DeduplicationHook
  g9_*   DGData.discretize
  g10_*  TGCN cell (gate wiring; GCNConv = placeholder restatement, third-party)
  g8_*   TGNMemory (in-tree arithmetic: messages, Last/Mean aggregation, GRU, store semantics)
  g11_*  NeighborSamplerHook (uniform sampling; Python's `random` seeded so the sampled rows are reproducible)
  g12_*  time-unit iteration: slice_time windows, DGDataLoader(batch_unit, drop_last, on_empty) incl. empty windows
  g13_*  the DGraph view surface: scalar properties, sparse ``node_x`` / ``node_y`` (indices, values, shape) and ``materialize()``
         over event / time / nested slices of a stream with node events, node labels, edge types and unsorted input
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, '_pyg_stub'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from tgm import DGraph  # noqa: E402
from tgm.data import DGData, DGDataLoader  # noqa: E402
from tgm.hooks import DeduplicationHook, HookManager, RecencyNeighborHook  # noqa: E402
from tgm.hooks.base import StatelessHook  # noqa: E402
from tgm.nn import TGAT, TemporalAttention, Time2Vec  # noqa: E402

from tgm_amd.synth import make_stream  # noqa: E402


class ReplayNegatives(StatelessHook):
    """Feeds pre-generated negatives (one per edge, indexed by global edge id)."""

    _cls_requires = {'edge_src', 'edge_dst', 'edge_time'}
    _cls_produces = {'neg', 'neg_time'}

    def __init__(self, neg_per_edge: torch.Tensor) -> None:
        super().__init__()
        self.neg_per_edge = neg_per_edge
        self.cursor = 0
        self.__post_init__()

    def __call__(self, dg, batch):
        n = batch.edge_src.numel()
        batch.neg = self.neg_per_edge[self.cursor : self.cursor + n].clone()
        batch.neg_time = batch.edge_time.clone()
        self.cursor += n
        return batch


def run_sampler(src, dst, ts, edge_x, num_nodes, num_nbrs, bs, directed, neg=None, plan=('epoch',), segments=None):
    """plan: sequence of 'epoch' | 'reset'.  segments: list of (lo, hi) edge ranges
    iterated in order inside each epoch (separate DGData each, like a train/val split)."""
    E = len(src)
    segments = segments or [(0, E)]
    seed_keys = ['edge_src', 'edge_dst'] + (['neg'] if neg is not None else [])
    time_keys = ['edge_time', 'edge_time'] + (['neg_time'] if neg is not None else [])
    hook = RecencyNeighborHook(
        num_nodes=num_nodes, num_nbrs=list(num_nbrs), seed_nodes_keys=seed_keys, seed_times_keys=time_keys, directed=directed
    )
    hm = HookManager(keys=['k'])
    replay = None
    if neg is not None:
        replay = ReplayNegatives(torch.as_tensor(neg, dtype=torch.int32))
        hm.register('k', replay)
    hm.register('k', hook)
    graphs = []
    for lo, hi in segments:
        d = DGData.from_raw(
            torch.as_tensor(ts[lo:hi], dtype=torch.int64),
            torch.stack([torch.as_tensor(src[lo:hi], dtype=torch.int32), torch.as_tensor(dst[lo:hi], dtype=torch.int32)], 1),
            None if edge_x is None else torch.as_tensor(edge_x[lo:hi], dtype=torch.float32),
        )
        graphs.append((lo, DGraph(d)))
    outs = []
    with hm.activate('k'):
        for step in plan:
            if step == 'reset':
                hm.reset_state()
                continue
            for lo, dg in graphs:
                if replay is not None:
                    replay.cursor = lo
                for batch in DGDataLoader(dg, batch_size=bs, hook_manager=hm):
                    hops = []
                    for h in range(len(num_nbrs)):
                        hops.append(
                            dict(
                                seed_nids=batch.seed_nids[h].numpy().copy(),
                                seed_times=batch.seed_times[h].numpy().copy(),
                                nbr_nids=batch.nbr_nids[h].numpy().copy(),
                                nbr_edge_time=batch.nbr_edge_time[h].numpy().copy(),
                                nbr_edge_x=batch.nbr_edge_x[h].numpy().copy(),
                            )
                        )
                    outs.append(hops)
    return outs


def save_sampler_case(name, src, dst, ts, edge_x, num_nodes, num_nbrs, bs, directed, neg=None, plan=('epoch',), segments=None, digest_only=False, verbatim=()):
    outs = run_sampler(src, dst, ts, edge_x, num_nodes, num_nbrs, bs, directed, neg, plan, segments)
    meta = dict(
        name=name, num_nodes=int(num_nodes), num_nbrs=list(num_nbrs), batch_size=int(bs), directed=bool(directed),
        has_neg=neg is not None, plan=list(plan), segments=[list(s) for s in (segments or [(0, len(src))])],
        num_batches=len(outs), digest_only=bool(digest_only),
    )  # fmt: skip
    arrays = dict(src=np.asarray(src, np.int32), dst=np.asarray(dst, np.int32), ts=np.asarray(ts, np.int64))
    if edge_x is not None and not digest_only:
        arrays['edge_x'] = np.asarray(edge_x, np.float32)
    if neg is not None:
        arrays['neg'] = np.asarray(neg, np.int32)
    digests = {}
    for b, hops in enumerate(outs):
        for h, d in enumerate(hops):
            for key, val in d.items():
                if digest_only:
                    digests[f'b{b}_h{h}_{key}'] = hashlib.sha256(np.ascontiguousarray(val).tobytes()).hexdigest()
                    if b in verbatim and key in ('nbr_nids', 'nbr_edge_time'):
                        arrays[f'b{b}_h{h}_{key}'] = val
                else:
                    arrays[f'b{b}_h{h}_{key}'] = val
    meta['digests'] = digests
    arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **arrays)
    print(f'{name}: {len(outs)} batches')


def g1_cases():
    # basic 4-edge graph
    src, dst, ts = [0, 0, 2, 2], [1, 2, 3, 0], [1, 2, 3, 4]
    x = np.array([[1], [2], [5], [2]], np.float32)
    save_sampler_case('g1_basic_1hop', src, dst, ts, x, 4, [1], 1, False)
    save_sampler_case('g1_basic_1hop_directed', src, dst, ts, x, 4, [1], 1, True)
    save_sampler_case('g1_basic_reset', src, dst, ts, x, 4, [1], 1, False, plan=('epoch', 'reset', 'epoch'))
    # star graph exceeding the buffer
    src, dst, ts = [0] * 100, list(range(1, 101)), list(range(100))
    x = np.arange(1, 101, dtype=np.float32).reshape(-1, 1)
    save_sampler_case('g1_buffer_k2', src, dst, ts, x, 101, [2], 2, False)
    # two-hop graph
    src, dst, ts = [0, 1, 3, 4, 5, 5], [1, 2, 2, 2, 0, 2], [1, 2, 3, 4, 5, 6]
    x = np.array([[1], [3], [5], [6], [5], [7]], np.float32)
    save_sampler_case('g1_two_hop', src, dst, ts, x, 6, [1, 1], 1, False)
    save_sampler_case('g1_two_hop_directed', src, dst, ts, x, 6, [1, 1], 1, True)
    # no edge features
    save_sampler_case('g1_no_edge_feat', [1, 2, 3], [2, 3, 4], [1, 2, 3], None, 5, [1], 3, True)


def g2_cases():
    rng = np.random.default_rng(20260928)
    variants = [([2], 3), ([3, 2], 4), ([2, 3], 5), ([1, 1, 2], 2), ([4], 7), ([3, 3], 1), ([5, 2], 8), ([2, 2, 2], 6)]
    for i in range(16):
        num_nbrs, bs = variants[i % len(variants)]
        N = int(rng.integers(3, 12))
        E = int(rng.integers(5, 80))
        src = rng.integers(0, N, E)
        dst = rng.integers(0, N, E)  # self loops & non-bipartite on purpose
        ts = np.sort(rng.integers(1, max(2, E // 3), E))  # heavy ties, all >= 1
        D = int(rng.integers(0, 4))
        x = rng.random((E, D), dtype=np.float32) if D else None
        directed = bool(i % 2)
        neg = rng.integers(0, N, E) if i % 3 == 0 else None
        plan = ('epoch', 'reset', 'epoch') if i % 5 == 0 else ('epoch',)
        save_sampler_case(f'g2_rand_{i:02d}', src, dst, ts, x, N, num_nbrs, bs, directed, neg=neg, plan=plan)


def g3_case():
    st = make_stream('wiki', seed=1337, num_edges=20_000, edge_dim=8)
    rng = np.random.default_rng(7)
    neg = rng.integers(8227, st.num_nodes, st.num_edges)
    save_sampler_case(
        'g3_wiki_medium', st.src.numpy(), st.dst.numpy(), st.ts.numpy(), st.edge_x.numpy(), st.num_nodes, [20, 20], 200, False,
        neg=neg, digest_only=True, verbatim=(0, 1, 50, 99),
    )  # fmt: skip


def g4_case():
    rng = np.random.default_rng(99)
    N, E, D = 40, 600, 3
    src, dst = rng.integers(0, N, E), rng.integers(0, N, E)
    ts = np.sort(rng.integers(1, 200, E))
    x = rng.random((E, D), dtype=np.float32)
    neg = rng.integers(0, N, E)
    # train [0,420) then val [420,600) sharing hook state; reset; second epoch
    save_sampler_case(
        'g4_epoch_carry', src, dst, ts, x, N, [4, 3], 32, False, neg=neg,
        plan=('epoch', 'reset', 'epoch'), segments=[(0, 420), (420, 600)],
    )  # fmt: skip


def _jitter_params(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g))


def g5_cases():
    torch.manual_seed(1337)
    # --- TemporalAttention alone, incl. all-masked rows and head padding (node_dim 1 -> pad 1)
    for tag, (heads, nd, ed, td, Bn, k) in {
        'pad': (2, 1, 6, 5, 7, 4),
        'nopad': (2, 4, 3, 6, 5, 3),
        'h4': (4, 7, 5, 9, 6, 5),
    }.items():
        m = TemporalAttention(n_heads=heads, node_dim=nd, edge_dim=ed, time_dim=td, dropout=0.1).eval()
        _jitter_params(m, 5)
        g = torch.Generator().manual_seed(11)
        node_x = torch.randn(Bn, nd, generator=g)
        time_feat = torch.randn(Bn, td, generator=g)
        edge_feat = torch.randn(Bn, k, ed, generator=g)
        nbr_node_feat = torch.randn(Bn, k, nd, generator=g)
        nbr_time_feat = torch.randn(Bn, k, td, generator=g)
        mask = torch.rand(Bn, k, generator=g) > 0.4
        mask[0] = False  # a row with no valid neighbor
        mask[1] = True
        with torch.no_grad():
            out = m(node_x=node_x, time_feat=time_feat, edge_feat=edge_feat, nbr_node_feat=nbr_node_feat,
                    nbr_time_feat=nbr_time_feat, valid_nbr_mask=mask)  # fmt: skip
        arrays = {f'w_{n}': p.detach().numpy() for n, p in m.state_dict().items()}
        arrays.update(node_x=node_x.numpy(), time_feat=time_feat.numpy(), edge_feat=edge_feat.numpy(),
                      nbr_node_feat=nbr_node_feat.numpy(), nbr_time_feat=nbr_time_feat.numpy(), mask=mask.numpy(), out=out.numpy())  # fmt: skip
        arrays['meta'] = np.frombuffer(json.dumps(dict(n_heads=heads, node_dim=nd, edge_dim=ed, time_dim=td)).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f'g5_attn_{tag}.npz'), **arrays)
        print(f'g5_attn_{tag}: out {tuple(out.shape)}')

    # --- full TGAT on reference sampler outputs
    cases = {
        # name: (stream kwargs, num_nbrs, bs, batch index to keep, TGAT dims)
        'small_unix': (dict(shape='review', num_edges=3000, edge_dim=6, n_src=150, n_dst=40), [4, 4], 16, 100,
                       dict(node_dim=1, time_dim=10, embed_dim=12, n_heads=2)),
        'small_nd8': (dict(shape='wiki', num_edges=3000, edge_dim=5, n_src=120, n_dst=30, node_dim=8), [3, 3], 10, 150,
                      dict(node_dim=8, time_dim=7, embed_dim=8, n_heads=2)),
        'one_layer': (dict(shape='wiki', num_edges=2000, edge_dim=4, n_src=80, n_dst=20), [5], 12, 80,
                      dict(node_dim=1, time_dim=6, embed_dim=10, n_heads=2)),
        'example_dims': (dict(shape='wiki', num_edges=4000, edge_dim=172, n_src=300, n_dst=60), [20, 20], 8, 300,
                         dict(node_dim=1, time_dim=100, embed_dim=172, n_heads=2)),
    }
    for name, (skw, num_nbrs, bs, keep, dims) in cases.items():
        st = make_stream(seed=4242, **skw)
        rng = np.random.default_rng(3)
        lo = st.num_nodes - skw['n_dst'] if skw['shape'] != 'comment' else 0
        neg = rng.integers(lo, st.num_nodes, st.num_edges)
        outs = run_sampler(st.src.numpy(), st.dst.numpy(), st.ts.numpy(), st.edge_x.numpy(), st.num_nodes, num_nbrs, bs, False, neg=neg)
        hops = outs[keep]
        enc = TGAT(edge_dim=st.edge_dim, num_layers=len(num_nbrs), dropout=0.1, **dims).eval()
        _jitter_params(enc, 17)
        T = lambda a: torch.from_numpy(a)
        with torch.no_grad():
            z = enc(
                st.node_x,
                [T(h['seed_nids']) for h in hops],
                [T(h['seed_times']) for h in hops],
                [T(h['nbr_nids']) for h in hops],
                [T(h['nbr_edge_x']) for h in hops],
                [T(h['nbr_edge_time']) for h in hops],
            )
        arrays = {f'w_{n}': p.detach().numpy() for n, p in enc.state_dict().items()}
        arrays['z'] = z.numpy()
        arrays['node_x'] = st.node_x.numpy()
        for h, d in enumerate(hops):
            for key in ('seed_nids', 'seed_times', 'nbr_nids', 'nbr_edge_time'):
                arrays[f'h{h}_{key}'] = d[key]
            if name != 'example_dims':
                arrays[f'h{h}_nbr_edge_x'] = d['nbr_edge_x']
        meta = dict(stream=skw, stream_seed=4242, num_nbrs=num_nbrs, batch_size=bs, batch_index=keep, neg_seed=3, neg_lo=int(lo),
                    dims=dims, edge_dim=st.edge_dim, nbr_edge_x_stored=name != 'example_dims')  # fmt: skip
        arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f'g5_tgat_{name}.npz'), **arrays)
        print(f'g5_tgat_{name}: z {tuple(z.shape)}  |z|max {z.abs().max():.3f}')


def g5_self_noise():
    """How far is the REFERENCE from itself on the g5 fixtures?  Replays every committed g5 fixture (same weights, same inputs) through the
    reference three ways -- the default thread count (must reproduce the stored output bit for bit), ONE thread (another summation order
    inside the BLAS calls), and float64 arithmetic (module.double(), Time2Vec kept in float32: its float32 rounding is part of the
    function at unix-scale time deltas) -- and stores, per fixture, the float32 run's distance to both.
    tests/test_tgat_gpu.py prints a parity error as a multiple of these next to its multiple of the bound.  Data only (a JSON file)."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import golden_util as gu

    def stats(a, b):
        a, b = a.double(), b.double()
        err = (a - b).abs()
        big = b.abs() >= 1e-2
        return dict(max_abs=float(err.max()), worst_multiple_of_bound=float((err / (1e-5 * b.abs().clamp(min=1.0))).max()),
                    worst_relative_where_ref_ge_1e_2=float((err[big] / b.abs()[big]).max()) if bool(big.any()) else 0.0)

    def double_but_time2vec(enc):
        """enc.double(), except that Time2Vec stays what the reference defines it to be -- cos of a FLOAT32 fma over a float32 time delta
        (at unix-scale deltas the float32 rounding of w * dt IS the function: in float64 the cosine's argument is another number) -- and
        hands its float32 result over as float64."""
        enc = enc.double()
        te = enc.time_encoder.float()
        fwd = te.forward
        te.forward = lambda x: fwd(x).double()
        return enc

    out = {}
    n_threads = torch.get_num_threads()
    for case in gu.ATTN_CASES:
        meta, a = gu.load(case)
        T = lambda k: torch.from_numpy(a[k])
        m = TemporalAttention(meta['n_heads'], meta['node_dim'], meta['edge_dim'], meta['time_dim'], dropout=0.1).eval()
        m.load_state_dict({k[2:]: torch.from_numpy(v) for k, v in a.items() if k.startswith('w_')})
        kw = dict(node_x=T('node_x'), time_feat=T('time_feat'), edge_feat=T('edge_feat'), nbr_node_feat=T('nbr_node_feat'),
                  nbr_time_feat=T('nbr_time_feat'), valid_nbr_mask=T('mask'))  # fmt: skip
        with torch.no_grad():
            z = m(**kw)
            torch.set_num_threads(1)
            z1 = m(**kw)
            torch.set_num_threads(n_threads)
            z64 = m.double()(**{k: (v.double() if v.dtype == torch.float32 else v) for k, v in kw.items()})
        out[case] = dict(reproduces_fixture=bool(torch.equal(z, T('out'))), threads=[n_threads, 1], one_thread_vs_default=stats(z1, z),
                         float32_vs_float64=stats(z, z64), max_abs_ref=float(z.abs().max()))
        print(case, out[case])
    for case in gu.TGAT_CASES:
        meta, params, inputs, z_ref = gu.tgat_case(case)
        enc = TGAT(edge_dim=meta['edge_dim'], num_layers=len(meta['num_nbrs']), dropout=0.1, **meta['dims']).eval()
        enc.load_state_dict(params)
        args = [inputs[k] for k in ('node_x', 'seed_nids', 'seed_times', 'nbr_nids', 'nbr_edge_x', 'nbr_edge_time')]
        with torch.no_grad():
            z = enc(*args)
            torch.set_num_threads(1)
            z1 = enc(*args)
            torch.set_num_threads(n_threads)
            d = lambda v: [d(x) for x in v] if isinstance(v, list) else (v.double() if v.dtype == torch.float32 else v)
            z64 = double_but_time2vec(enc)(*[d(v) for v in args])
        out[case] = dict(reproduces_fixture=bool(torch.equal(z, z_ref)), threads=[n_threads, 1], one_thread_vs_default=stats(z1, z),
                         float32_vs_float64=stats(z, z64), max_abs_ref=float(z.abs().max()))
        print(case, out[case])
    out['_what'] = ('the reference (tgm.nn TemporalAttention / TGAT, eval mode, torch-CPU float32) against ITSELF on the g5 fixtures: one BLAS thread vs the '
                    'default, and float32 vs float64 arithmetic; max_abs = max |a - b|, worst_multiple_of_bound = max |a - b| / (1e-5 max(1, |b|)), '
                    'worst_relative_where_ref_ge_1e_2 = max |a - b| / |b| over |b| >= 1e-2.  Written by tests/golden/make_golden.py g5n in the build container.')
    with open(os.path.join(HERE, 'g5_self_noise.json'), 'w') as f:
        json.dump(out, f, indent=1)


def g6_case():
    enc = Time2Vec(time_dim=100).eval()
    t = torch.tensor(
        [0, 1, 2, 3, 7, 59, 60, 3600, 86_399, 86_400, 604_800, 2_678_373, 16_777_216, 16_777_217, 123_456_789,
         999_999_937, 1_500_000_000, 2_147_483_646, 2_147_483_647, 2_147_483_648],
        dtype=torch.int64,
    )  # fmt: skip
    with torch.no_grad():
        out = enc(t)
    enc2 = Time2Vec(time_dim=16).eval()
    _jitter_params(enc2, 23)
    with torch.no_grad():
        out2 = enc2(t)
    np.savez_compressed(
        os.path.join(HERE, 'g6_time2vec.npz'), t=t.numpy(), out_default=out.numpy(), w2=enc2.w.weight.detach().numpy(),
        b2=enc2.w.bias.detach().numpy(), out_jitter=out2.numpy(),
    )  # fmt: skip
    print('g6_time2vec:', tuple(out.shape))


def g7_case():
    rng = np.random.default_rng(5)
    N, E = 30, 200
    src, dst = rng.integers(0, N, E), rng.integers(0, N, E)
    ts = np.sort(rng.integers(1, 100, E))
    neg = rng.integers(0, N, E)
    d = DGData.from_raw(torch.as_tensor(ts), torch.stack([torch.as_tensor(src, dtype=torch.int32), torch.as_tensor(dst, dtype=torch.int32)], 1))
    dg = DGraph(d)
    hm = HookManager(keys=['k'])
    hm.register('k', ReplayNegatives(torch.as_tensor(neg, dtype=torch.int32)))
    hm.register('k', RecencyNeighborHook(num_nodes=N, num_nbrs=[3, 2], seed_nodes_keys=['edge_src', 'edge_dst', 'neg'],
                                        seed_times_keys=['edge_time', 'edge_time', 'neg_time']))  # fmt: skip
    hm.register('k', DeduplicationHook(seed_nodes_keys=['neg', 'nbr_nids']))
    arrays = dict(src=src.astype(np.int32), dst=dst.astype(np.int32), ts=ts.astype(np.int64), neg=neg.astype(np.int32))
    nb = 0
    with hm.activate('k'):
        for b, batch in enumerate(DGDataLoader(dg, batch_size=25, hook_manager=hm)):
            arrays[f'b{b}_unique_nids'] = batch.unique_nids.numpy().copy()
            arrays[f'b{b}_local_src'] = batch.global_to_local(batch.edge_src).numpy().copy()
            nb += 1
    arrays['meta'] = np.frombuffer(json.dumps(dict(num_nodes=N, num_nbrs=[3, 2], batch_size=25, num_batches=nb)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'g7_dedup.npz'), **arrays)
    print('g7_dedup:', nb, 'batches')


def g8_cases():
    """TGN memory module, in-tree arithmetic only (tgm/nn/encoder/tgn.py:43-251): train-mode lookahead
    `memory(n_id)`, `update_state`, the train->eval flush and eval-mode updates, for Last / Mean aggregation.
    Streams have no two events of one node with equal float32 time inside a batch: the reference orders a node's
    stored events with a NON-stable sort (tgn.py:226), so tie-breaking there is unspecified."""
    from tgm.nn.encoder.tgn import IdentityMessage, LastAggregator, MeanAggregator, TGNMemory

    for tag, aggr_cls, base, step in (('last', LastAggregator, 0, 3), ('mean', MeanAggregator, 0, 3), ('last_unix', LastAggregator, 1_200_000_000, 700)):
        rng = np.random.default_rng(31)
        N, E, D, M, T, bs = 40, 400, 5, 8, 6, 16
        src = rng.integers(0, N, E).astype(np.int32)
        dst = rng.integers(0, N, E).astype(np.int32)
        ts = (base + np.cumsum(rng.integers(1, step + 1, E) * (256 if base else 1))).astype(np.int64)
        raw = rng.random((E, D), dtype=np.float32)
        neg = rng.integers(0, N, E).astype(np.int32)
        torch.manual_seed(7)
        mem = TGNMemory(N, D, M, T, message_module=IdentityMessage(D, M, T), aggregator_module=aggr_cls())
        _jitter_params(mem, 41)
        arrays = dict(src=src, dst=dst, ts=ts, raw=raw, neg=neg)
        arrays.update({f'w_{n}': p.detach().numpy().copy() for n, p in mem.state_dict().items() if n not in ('memory', 'last_update', '_assoc')})
        T_ = torch.from_numpy
        nb_train = 18
        mem.train()
        b = 0
        with torch.no_grad():
            for lo in range(0, E, bs):
                hi = min(lo + bs, E)
                if b == nb_train:
                    mem.eval()  # flushes the message store into the memory (tgn.py:245-251)
                    arrays['flush_memory'] = mem.memory.numpy().copy()
                    arrays['flush_last_update'] = mem.last_update.numpy().copy()
                n_id = torch.unique(torch.cat([T_(src[lo:hi]), T_(dst[lo:hi]), T_(neg[lo:hi])])).long()
                z, lu = mem(n_id)
                arrays[f'b{b}_n_id'] = n_id.numpy().copy()
                arrays[f'b{b}_z'] = z.numpy().copy()
                arrays[f'b{b}_last_update'] = lu.numpy().copy()
                mem.update_state(T_(src[lo:hi]).long(), T_(dst[lo:hi]).long(), T_(ts[lo:hi]), T_(raw[lo:hi]))
                if b % 5 == 4 or b >= nb_train:
                    arrays[f'b{b}_memory'] = mem.memory.numpy().copy()
                    arrays[f'b{b}_mem_last_update'] = mem.last_update.numpy().copy()
                b += 1
        meta = dict(num_nodes=N, raw_msg_dim=D, memory_dim=M, time_dim=T, batch_size=bs, aggr=tag.split('_')[0], num_batches=b, train_batches=nb_train)
        arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f'g8_tgn_{tag}.npz'), **arrays)
        print(f'g8_tgn_{tag}: {b} batches')


def g9_case():
    """DGData.discretize (tgm/data/dg_data.py:423-564): seconds -> minutes / hours, with node events and labels."""
    rng = np.random.default_rng(77)
    E, N = 500, 12
    ts = np.sort(rng.integers(0, 20_000, E)).astype(np.int64)
    ei = rng.integers(0, N, (E, 2)).astype(np.int32)
    ex = rng.random((E, 3), dtype=np.float32)
    nt = np.sort(rng.integers(0, 20_000, 60)).astype(np.int64)
    nn_ = rng.integers(0, N, 60).astype(np.int32)
    nx = rng.random((60, 2), dtype=np.float32)
    yt = np.sort(rng.integers(0, 20_000, 40)).astype(np.int64)
    yn = rng.integers(0, N, 40).astype(np.int32)
    yv = rng.random((40, 4), dtype=np.float32)
    arrays = dict(ts=ts, ei=ei, ex=ex, nt=nt, nn=nn_, nx=nx, yt=yt, yn=yn, yv=yv)
    T_ = torch.from_numpy
    d = DGData.from_raw(T_(ts), T_(ei), T_(ex), node_x_time=T_(nt), node_x_nids=T_(nn_), node_x=T_(nx), node_y_time=T_(yt),
                        node_y_nids=T_(yn), node_y=T_(yv), time_delta='s')  # fmt: skip
    for unit in ('m', 'h'):
        c = d.discretize(unit)
        for f in ('time', 'edge_mask', 'edge_index', 'edge_x', 'node_x_mask', 'node_x_nids', 'node_x', 'node_y_mask', 'node_y_nids', 'node_y'):
            arrays[f'{unit}_{f}'] = getattr(c, f).numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'g9_discretize.npz'), **arrays)
    print('g9_discretize: ok')


def g10_case():
    """TGCN cell (tgm/nn/encoder/tgcn.py): three snapshots with a carried hidden state; GCNConv is the
    placeholder restatement (third-party arithmetic), so this pins the reference's gate wiring."""
    from tgm.nn.encoder.tgcn import TGCN

    rng = np.random.default_rng(3)
    N, Fin, C = 37, 5, 8
    arrays = {}
    for tag, improved in (('plain', False), ('improved', True)):
        torch.manual_seed(2)
        cell = TGCN(Fin, C, improved=improved).eval()
        _jitter_params(cell, 13)
        arrays.update({f'{tag}_w_{n}': p.detach().numpy().copy() for n, p in cell.state_dict().items()})
        H = None
        with torch.no_grad():
            for snap in range(3):
                E = int(rng.integers(60, 120))
                ei = rng.integers(0, N, (2, E)).astype(np.int64)
                ei[:, :4] = ei[0, :4]  # a few explicit self loops
                ew = rng.random(E, dtype=np.float32) + 0.5 if snap == 2 else None
                x = rng.standard_normal((N, Fin)).astype(np.float32)
                H = cell(torch.from_numpy(x), torch.from_numpy(ei), None if ew is None else torch.from_numpy(ew), H)
                arrays[f'{tag}_s{snap}_x'], arrays[f'{tag}_s{snap}_ei'] = x, ei
                if ew is not None:
                    arrays[f'{tag}_s{snap}_ew'] = ew
                arrays[f'{tag}_s{snap}_H'] = H.numpy().copy()
    arrays['meta'] = np.frombuffer(json.dumps(dict(N=N, Fin=Fin, C=C)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'g10_tgcn.npz'), **arrays)
    print('g10_tgcn: ok')


def g11_cases():
    """Uniform neighbor sampler (tgm/hooks/neighbors/uniform.py + array_backend.get_nbrs).  `sparse`: every node has at
    most k candidates (deterministic: all neighbors, event order, left aligned).  `dense`: hubs with more candidates than
    k -- the reference calls random.sample, so the generator seeds Python's `random` once before the loader loop."""
    import random

    from tgm.hooks import NeighborSamplerHook

    rng = np.random.default_rng(11)
    cases = {
        'sparse': dict(N=60, E=90, k=[6, 4], bs=9, directed=False, D=3, tmax=40),
        'dense': dict(N=12, E=160, k=[4, 3], bs=16, directed=False, D=2, tmax=60),
        'dense_directed': dict(N=10, E=120, k=[5], bs=12, directed=True, D=0, tmax=30),
    }
    for tag, c in cases.items():
        N, E = c['N'], c['E']
        src = rng.integers(0, N, E).astype(np.int32)
        dst = rng.integers(0, N, E).astype(np.int32)
        ts = np.sort(rng.integers(1, c['tmax'], E)).astype(np.int64)  # heavy ties
        edge_x = rng.random((E, c['D']), dtype=np.float32) if c['D'] else None
        neg = rng.integers(0, N, E).astype(np.int32)
        hook = NeighborSamplerHook(num_nbrs=list(c['k']), seed_nodes_keys=['edge_src', 'edge_dst', 'neg'],
                                   seed_times_keys=['edge_time', 'edge_time', 'neg_time'], directed=c['directed'])  # fmt: skip
        hm = HookManager(keys=['k'])
        hm.register('k', ReplayNegatives(torch.as_tensor(neg)))
        hm.register('k', hook)
        d = DGData.from_raw(torch.as_tensor(ts), torch.stack([torch.as_tensor(src), torch.as_tensor(dst)], 1),
                            None if edge_x is None else torch.as_tensor(edge_x))  # fmt: skip
        dg = DGraph(d)
        arrays = dict(src=src, dst=dst, ts=ts, neg=neg)
        if edge_x is not None:
            arrays['edge_x'] = edge_x
        random.seed(4242)
        nb = 0
        with hm.activate('k'):
            for b, batch in enumerate(DGDataLoader(dg, batch_size=c['bs'], hook_manager=hm)):
                nb += 1
                for h in range(len(c['k'])):
                    arrays[f'b{b}_h{h}_seed_nids'] = batch.seed_nids[h].numpy().copy()
                    arrays[f'b{b}_h{h}_nbr_nids'] = batch.nbr_nids[h].numpy().copy()
                    arrays[f'b{b}_h{h}_nbr_edge_time'] = batch.nbr_edge_time[h].numpy().copy()
                    arrays[f'b{b}_h{h}_nbr_edge_x'] = batch.nbr_edge_x[h].numpy().copy()
        meta = dict(name=f'g11_uniform_{tag}', num_nodes=N, num_nbrs=list(c['k']), batch_size=c['bs'], directed=c['directed'],
                    num_batches=nb, edge_dim=c['D'], random_seed=4242)  # fmt: skip
        arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f'g11_uniform_{tag}.npz'), **arrays)
        print(f'g11_uniform_{tag}: {nb} batches')


def g12_case():
    """Time-unit iteration (tgm/data/loader.py:101-170, tgm/core/graph.py:130-152, array_backend.py:301-321): ``slice_time`` windows and
    ``DGDataLoader(batch_unit=..., batch_size=..., drop_last=..., on_empty=...)`` over a seconds-granularity stream with long silent
    gaps (empty windows).  Per loader configuration: the number of batches, every batch's edges, or the exception the iteration ends in."""
    from tgm.exceptions import EmptyBatchError

    rng = np.random.default_rng(1212)
    N, D = 40, 3
    # three bursts of activity with silent gaps between them (so minute / hour windows come up empty), ties inside the bursts
    parts = [np.sort(rng.integers(lo, hi, n)) for lo, hi, n in ((5, 400, 90), (2_000, 2_300, 60), (9_000, 12_700, 150))]
    ts = np.concatenate(parts).astype(np.int64)
    E = len(ts)
    ei = rng.integers(0, N, (E, 2)).astype(np.int32)
    ex = rng.random((E, D), dtype=np.float32)
    arrays = dict(ts=ts, ei=ei, ex=ex)
    T_ = torch.from_numpy
    dg = DGraph(DGData.from_raw(T_(ts), T_(ei), T_(ex), time_delta='s'))
    meta = dict(N=N, start_time=int(dg.start_time), end_time=int(dg.end_time), slices=[], loaders=[], x_none=[])

    def put(tag, b):
        arrays[f'{tag}_src'] = b.edge_src.numpy().copy()
        arrays[f'{tag}_dst'] = b.edge_dst.numpy().copy()
        arrays[f'{tag}_time'] = b.edge_time.numpy().copy()
        if b.edge_x is None:  # (the reference materializes no feature tensor for some empty slices: recorded, compared as such)
            meta['x_none'].append(tag)
        else:
            arrays[f'{tag}_x'] = b.edge_x.numpy().copy()

    # slice_time: half-open [start, end) in the graph's unit, None = open, nested slices intersect
    windows = [(None, None), (0, 5), (5, 6), (0, 401), (400, 2_000), (2_000, None), (None, 9_000), (12_699, 12_700), (12_700, 20_000), (100, 100)]
    for i, (a, b) in enumerate(windows):
        put(f'sl{i}', dg.slice_time(a, b).materialize())
        meta['slices'].append([a, b])
    nested = dg.slice_time(50, 11_000).slice_time(None, 2_100).slice_events(3, None)
    put('nested', nested.materialize())
    meta['nested'] = dict(num_events=int(nested.num_events), start_time=int(nested.start_time), end_time=int(nested.end_time))

    configs = []
    for unit, size in (('s', 250), ('s', 3_600), ('m', 1), ('m', 7), ('h', 1), ('h', 2)):
        for drop_last in (False, True):
            for on_empty in ('skip', 'raise', None):
                configs.append((unit, size, drop_last, on_empty))
    for ci, (unit, size, drop_last, on_empty) in enumerate(configs):
        rec = dict(unit=unit, size=size, drop_last=drop_last, on_empty=on_empty, error=None)
        loader = DGDataLoader(dg, batch_size=size, batch_unit=unit, on_empty=on_empty, drop_last=drop_last)
        rec['len'] = len(loader)
        n = 0
        got = dict(sizes=[], src=[], dst=[], time=[], x=[])
        try:
            for b in loader:
                got['sizes'].append(b.edge_src.numel())
                got['src'].append(b.edge_src.numpy().copy()); got['dst'].append(b.edge_dst.numpy().copy()); got['time'].append(b.edge_time.numpy().copy())
                got['x'].append(np.zeros((0, D), np.float32) if b.edge_x is None else b.edge_x.numpy().copy())
                if b.edge_x is None:
                    meta['x_none'].append(f'c{ci}_b{n}')
                n += 1
        except EmptyBatchError:
            rec['error'] = 'EmptyBatchError'
        # one record per configuration: the batch sizes + the batches' edges concatenated in iteration order
        arrays[f'c{ci}_sizes'] = np.asarray(got['sizes'], np.int64)
        for f, dt in (('src', np.int32), ('dst', np.int32), ('time', np.int64)):
            arrays[f'c{ci}_{f}'] = np.concatenate(got[f]).astype(dt) if got[f] else np.zeros(0, dt)
        if on_empty == 'skip':  # (the feature rows once per (unit, size, drop_last): the other on_empty modes iterate the same slices)
            arrays[f'c{ci}_x'] = np.concatenate(got['x']) if got['x'] else np.zeros((0, D), np.float32)
        rec['batches'] = n
        meta['loaders'].append(rec)
    arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'g12_time_batches.npz'), **arrays)
    print(f'g12_time_batches: {len(windows)} slices, {len(configs)} loader configurations, E={E}')


def g13_case():
    """``DGraph`` as a script reads it (tgm/core/graph.py:74-108, 186-356; array_backend.py:178-285): per slice the scalar properties,
    ``dg.node_x`` / ``dg.node_y`` (``sparse_coo_tensor(T x V x d)``: indices, values, shape -- or None) and every ``materialize()`` field.
    The stream mixes edges, dynamic node features and node labels with timestamp ties between the three kinds, and is handed to
    ``DGData.from_raw`` unsorted."""
    rng = np.random.default_rng(1313)
    N, E, NX, NY, D, DX, DY = 23, 60, 25, 18, 3, 4, 2
    T_ = torch.from_numpy
    perm = rng.permutation(E)
    ets = np.sort(rng.integers(3, 90, E)).astype(np.int64)[perm]
    ei = rng.integers(0, N - 4, (E, 2)).astype(np.int32)  # (ids N-4.. appear only as node events / labels: they move the sparse shapes)
    ex = rng.random((E, D), dtype=np.float32)
    et = rng.integers(0, 5, E).astype(np.int32)
    xts = rng.integers(0, 100, NX).astype(np.int64)
    xid = rng.integers(0, N, NX).astype(np.int32)
    xid[0] = N - 1  # (the labels' ids must stay inside the id range the edges and node events span)
    xv = rng.random((NX, DX), dtype=np.float32)
    yts = rng.integers(0, 100, NY).astype(np.int64)
    yid = rng.integers(0, N, NY).astype(np.int32)
    yv = rng.random((NY, DY), dtype=np.float32)
    sx = rng.random((N, 6), dtype=np.float32)
    nt = rng.integers(0, 3, N).astype(np.int32)
    arrays = dict(ets=ets, ei=ei, ex=ex, et=et, xts=xts, xid=xid, xv=xv, yts=yts, yid=yid, yv=yv, sx=sx, nt=nt)
    data = DGData.from_raw(T_(ets), T_(ei), T_(ex), T_(xts), T_(xid), T_(xv), T_(yts), T_(yid), T_(yv), static_node_x=T_(sx), edge_type=T_(et),
                           node_type=T_(nt))  # fmt: skip
    # each entry: a chain of ('t', start, end) / ('e', start, end) slice operations applied to the full graph
    chains = [[], [['t', 10, 50]], [['t', None, 30]], [['t', 60, None]], [['e', 5, 40]], [['e', None, 17]], [['e', 70, None]], [['e', 20, 80], ['t', 30, 70]],
              [['t', 20, 80], ['e', 30, 60], ['t', 40, None]], [['t', 95, 99]], [['t', 0, 3]], [['e', 0, 1]], [['t', 200, 300]], [['e', 50, 50]]]  # fmt: skip
    meta = dict(chains=chains, views=[])
    for i, chain in enumerate(chains):
        dg = DGraph(data)
        for kind, a, b in chain:
            dg = dg.slice_time(a, b) if kind == 't' else dg.slice_events(a, b)
        # (read like a script would: the batch first, the view's properties afterwards)
        batch = dg.materialize()
        rec = dict(len=len(dg), num_nodes=int(dg.num_nodes), num_node_events=int(dg.num_node_events), num_node_labels=int(dg.num_node_labels),
                   num_edge_events=int(dg.num_edge_events), num_timestamps=int(dg.num_timestamps), num_events=int(dg.num_events),
                   start_time=None if dg.start_time is None else int(dg.start_time), end_time=None if dg.end_time is None else int(dg.end_time),
                   none=[])  # fmt: skip
        for name in ('node_x', 'node_y'):
            sp = getattr(dg, name)
            if sp is None:
                rec['none'].append(name)
                continue
            arrays[f'v{i}_{name}_indices'] = sp._indices().numpy().copy()
            arrays[f'v{i}_{name}_values'] = sp._values().numpy().copy()
            rec[f'{name}_shape'] = [int(v) for v in sp.shape]
        for name in ('edge_src', 'edge_dst', 'edge_time', 'edge_x', 'edge_type', 'node_x_time', 'node_x_nids', 'node_x', 'node_y_time', 'node_y_nids', 'node_y'):
            v = getattr(batch, name)
            if v is None:
                rec['none'].append('batch.' + name)
            else:
                arrays[f'v{i}_b_{name}'] = v.numpy().copy()
        for name in ('node_x_nids', 'node_x_time', 'node_y_nids', 'node_y_time'):
            arrays[f'v{i}_{name}'] = getattr(dg, name).numpy().copy()
        meta['views'].append(rec)
    arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'g13_dgraph_views.npz'), **arrays)
    print(f'g13_dgraph_views: {len(chains)} views, E={E}, node events {NX}, node labels {NY}')


if __name__ == '__main__':
    import warnings

    warnings.filterwarnings('ignore')
    only = sys.argv[1:]
    for fam, fn in [('g1', g1_cases), ('g2', g2_cases), ('g3', g3_case), ('g4', g4_case), ('g5', g5_cases), ('g5n', g5_self_noise), ('g6', g6_case), ('g7', g7_case), ('g8', g8_cases), ('g9', g9_case), ('g10', g10_case), ('g11', g11_cases), ('g12', g12_case), ('g13', g13_case)]:
        if not only or fam in only:
            fn()

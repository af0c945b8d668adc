"""Strong-scaling edge-batch sharding, two ranks (sharing the one test GPU; gloo for the rendezvous -- there is NO collective on
the sampler's data path): rank r samples the edges [n r / 2, n (r + 1) / 2) of every global batch.
Put back in the single-rank row order, the ranks' tensors must equal the single-rank tensors bit for bit -- ids, times,
feature rows AND the generated negatives (a rank's draws are a slice of the whole batch's: tgmx_recency_step_t.neg_index0) --
in the stateless mode (static index: DESIGN.md's multi-GPU mode) and with streaming rings (every rank replays the whole
batch's update), through the lowered chain and hook by hook.  On a multi-GPU node the same code runs one rank per GPU."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

BS, KS, E, D = 200, [6, 4], 2300, 8  # 2300 % 200 = 100: a ragged last batch; odd shares exist (bs 200 / 2 ranks is even, the last is 50)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, out, mode, pool):
    import torch.distributed as dist

    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.dist import EdgeShardHook
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook
    from tgm_amd.synth import make_stream

    if world > 1:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    st = make_stream('comment', seed=9, num_edges=E, edge_dim=D, n_src=300)
    dg = DGraph(DGData.from_raw(st.ts, torch.stack([st.src, st.dst], 1), st.edge_x), device='cuda')
    hm = HookManager(keys=['k'])
    if world > 1:
        hm.register('k', EdgeShardHook(rank, world))
        hm.register('k', RandomNegativeEdgeSamplerHook(0, st.num_nodes, seed=17, like='shard_dst', time_key='shard_time'))
        keys, tkeys = ['shard_src', 'shard_dst', 'neg'], ['shard_time', 'shard_time', 'neg_time']
    else:
        hm.register('k', RandomNegativeEdgeSamplerHook(0, st.num_nodes, seed=17))
        keys, tkeys = ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time']
    hm.register('k', RecencyNeighborHook(st.num_nodes, KS, keys, tkeys, mode=mode, validate='deferred', key_arith='int64',
                                         batch_size=BS if mode == 'csr' else None))  # fmt: skip
    loader = DGDataLoader(dg, batch_size=BS, hook_manager=hm, output_pool=pool)
    got = []
    with hm.activate('k'):
        for b in loader:
            got.append({'neg': b.neg.cpu(), 'seed_nids': [t.cpu() for t in b.seed_nids], 'nbr_nids': [t.cpu() for t in b.nbr_nids],
                        'nbr_edge_time': [t.cpu() for t in b.nbr_edge_time], 'nbr_edge_x': [t.cpu() for t in b.nbr_edge_x]})
    torch.save(got, f'{out}.{mode}.{pool}.{world}.{rank}')
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _rows(share_sizes, roles, k_prod):
    """Single-rank row order of hop h (k_prod = rows per hop-0 seed) from the ranks' [role-major per rank] rows: for every role,
    rank 0's share then rank 1's."""
    order = []
    for role in range(roles):
        for r, n in enumerate(share_sizes):
            order.append((r, role * n * k_prod, n * k_prod))
    return order


@pytest.mark.timeout(600)
@pytest.mark.parametrize('mode,pool', [('csr', None), ('ring', None), ('ring', 0)])
def test_two_rank_strong_scaling_concat_equals_single_rank(tmp_path, mode, pool):
    out = str(tmp_path / 'o')
    _run(0, 1, 0, out, mode, pool)
    mp.spawn(_run, args=(2, _free_port(), out, mode, pool), nprocs=2, join=True)
    one = torch.load(f'{out}.{mode}.{pool}.1.0')
    two = [torch.load(f'{out}.{mode}.{pool}.2.{r}') for r in (0, 1)]
    assert len(one) == len(two[0]) == len(two[1]) == -(-E // BS)
    n_valid = 0
    for b, ref in enumerate(one):
        parts = [two[0][b], two[1][b]]
        shares = [p['neg'].shape[0] for p in parts]
        assert sum(shares) == ref['neg'].shape[0]
        assert torch.equal(torch.cat([p['neg'] for p in parts]), ref['neg']), f'batch {b}: the shares\' negatives are not slices of the batch\'s'
        k_prod = 1
        for h in range(len(KS)):
            for name in ('seed_nids', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x'):
                rebuilt = torch.cat([parts[r][name][h][lo:lo + n] for r, lo, n in _rows(shares, 3, k_prod)])
                assert torch.equal(rebuilt, ref[name][h]), f'{mode} pool={pool} batch {b} hop {h} {name}'
            n_valid += int((ref['nbr_nids'][h] >= 0).sum())
            k_prod *= KS[h]
    assert n_valid > 1000


# ---- batch-level sharding (round 4): rank r takes batches r, r + world, ... of the SAME schedule -------------------------------------
def _tie_stream(num_edges=E, n=40, seed=5):
    """Non-bipartite, few nodes, every timestamp shared by ~37 consecutive events: runs of equal time cross the bs = 200 boundaries and
    most nodes appear in both roles inside a run -- the case in which the static index's order depends on where the batches start."""
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (num_edges,), generator=g, dtype=torch.int32)
    dst = torch.randint(0, n, (num_edges,), generator=g, dtype=torch.int32)
    ts = (torch.arange(num_edges, dtype=torch.int64) // 37) * 10
    x = torch.randn(num_edges, D, generator=g)
    return src, dst, ts, x, n


def _run_batches(rank, world, port, out, pool, epochs, ties=False):
    import torch.distributed as dist

    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook
    from tgm_amd.synth import make_stream

    if world > 1:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    if ties:
        src, dst, ts, x, num_nodes = _tie_stream()
    else:
        st = make_stream('comment', seed=9, num_edges=E, edge_dim=D, n_src=300)
        src, dst, ts, x, num_nodes = st.src, st.dst, st.ts, st.edge_x, st.num_nodes
    dg = DGraph(DGData.from_raw(ts, torch.stack([src, dst], 1), x), device='cuda')
    hm = HookManager(keys=['k'])
    hm.register('k', RandomNegativeEdgeSamplerHook(0, num_nodes, seed=17))
    hm.register('k', RecencyNeighborHook(num_nodes, KS, ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time'], mode='csr',
                                         batch_size=BS))  # fmt: skip
    loader = DGDataLoader(dg, batch_size=BS, hook_manager=hm, output_pool=pool, batch_shard=(rank, world) if world > 1 else None)
    got = []
    with hm.activate('k'):
        for ep in range(epochs):
            hm.reset_state()
            for b in loader:
                got.append({'lo': int(b._edge_lo), 'neg': b.neg.cpu(), 'seed_nids': [t.cpu() for t in b.seed_nids], 'nbr_nids': [t.cpu() for t in b.nbr_nids],
                            'nbr_edge_time': [t.cpu() for t in b.nbr_edge_time], 'nbr_edge_x': [t.cpu() for t in b.nbr_edge_x]})
    torch.save(got, f'{out}.b.{pool}.{world}.{rank}')
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('pool', [None, 0])
def test_three_rank_batch_sharding_with_time_ties_across_batch_boundaries(tmp_path, pool):
    """world = 3 on a tie-heavy non-bipartite stream: rank 2's first batch is batch 2, and an index whose leading batch merged batches
    0 and 1 would order an equal-time (src role, dst role) pair that straddles the 200-edge boundary differently from the single-process
    index (csr.hip clamps equal-time runs to a batch).  The index is built from the SCHEDULE's first edge on every rank."""
    out = str(tmp_path / 'o')
    _run_batches(0, 1, 0, out, pool, 1, True)
    mp.spawn(_run_batches, args=(3, _free_port(), out, pool, 1, True), nprocs=3, join=True)
    one = torch.load(f'{out}.b.{pool}.1.0')
    three = [torch.load(f'{out}.b.{pool}.3.{r}') for r in range(3)]
    nb = -(-E // BS)
    assert len(one) == nb and sum(len(t) for t in three) == nb
    n_valid = 0
    for j, ref in enumerate(one):
        got = three[j % 3][j // 3]
        assert got['lo'] == ref['lo'] == j * BS
        for h in range(len(KS)):
            for name in ('seed_nids', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x'):
                assert torch.equal(got[name][h], ref[name][h]), f'batch {j} (rank {j % 3}) hop {h} {name}'
            n_valid += int((ref['nbr_nids'][h] >= 0).sum())
    assert n_valid > 2000


@pytest.mark.timeout(600)
@pytest.mark.parametrize('pool', [None, 0])
def test_two_rank_batch_sharding_interleaves_to_the_single_rank_sequence(tmp_path, pool):
    """DGDataLoader(batch_shard=(rank, 2)) over the static index: every batch a rank produces IS the batch the single-rank loader
    produces at that position of the bs = 200 schedule -- same edges, same generated negatives, same sampled neighbors, bit for
    bit, over two epochs (the epoch anchor and the negative sampler's call counter carry over) -- so the cfg-4 workload at its OWN
    batch size spreads over G ranks with no exchange at all."""
    out = str(tmp_path / 'o')
    _run_batches(0, 1, 0, out, pool, 2)
    mp.spawn(_run_batches, args=(2, _free_port(), out, pool, 2), nprocs=2, join=True)
    one = torch.load(f'{out}.b.{pool}.1.0')
    two = [torch.load(f'{out}.b.{pool}.2.{r}') for r in (0, 1)]
    nb = -(-E // BS)
    assert len(one) == 2 * nb and len(two[0]) + len(two[1]) == 2 * nb
    n_valid = 0
    for ep in range(2):
        for j in range(nb):
            ref = one[ep * nb + j]
            per_rank = [len(range(r, nb, 2)) for r in (0, 1)]
            got = two[j % 2][ep * per_rank[j % 2] + j // 2]
            assert got['lo'] == ref['lo'] == j * BS
            assert torch.equal(got['neg'], ref['neg']), f'epoch {ep} batch {j}: negatives'
            for h in range(len(KS)):
                for name in ('seed_nids', 'nbr_nids', 'nbr_edge_time', 'nbr_edge_x'):
                    assert torch.equal(got[name][h], ref[name][h]), f'epoch {ep} batch {j} hop {h} {name}'
                n_valid += int((ref['nbr_nids'][h] >= 0).sum())
    assert n_valid > 2000

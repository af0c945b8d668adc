"""The N>1 path on CPU: two gloo ranks, edge-batch sharding and the max-over-ranks timing reduction
(no kernels run here -- the sharding arithmetic and the process-group plumbing are what is under test)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.dist import EdgeShardHook, init_process_group
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook

    r, w, _ = init_process_group('gloo')
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)  # replicated stream: every rank builds the same one
    E = 103
    ts = torch.sort(torch.randint(0, 50, (E,), generator=g)).values
    ei = torch.randint(0, 11, (E, 2), generator=g, dtype=torch.int32)
    dg = DGraph(DGData.from_raw(ts, ei, torch.rand(E, 2, generator=g)))
    hm = HookManager(keys=['k'])
    hm.register('k', EdgeShardHook(rank, world))
    hm.register('k', RandomNegativeEdgeSamplerHook(0, 11, like='shard_dst', time_key='shard_time'))
    pieces = []
    with hm.activate('k'):
        for b in DGDataLoader(dg, batch_size=10 * world, hook_manager=hm):
            assert b.neg.shape == b.shard_dst.shape and torch.equal(b.neg_time, b.shard_time)
            pieces.append(torch.stack([b.shard_src, b.shard_dst]).clone())
    mine = torch.cat(pieces, dim=1)
    # gather every rank's shards and check that, batch by batch in rank order, they tile the stream exactly
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine.shape[1]]))
    width = max(int(s) for s in sizes)  # gloo all_gather wants equal shapes: pad to the widest shard
    padded = torch.zeros((2, width), dtype=torch.int32)
    padded[:, : mine.shape[1]] = mine
    got = [torch.zeros((2, width), dtype=torch.int32) for _ in sizes]
    dist.all_gather(got, padded)
    got = [g[:, : int(s)] for g, s in zip(got, sizes)]
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's "max over ranks" reduction
    if rank == 0:
        n_per = [0] * world
        rebuilt = []
        for s in range(0, E, 10 * world):
            n = min(10 * world, E - s)
            for q in range(world):
                lo, hi = (n * q) // world, (n * (q + 1)) // world
                rebuilt.append(got[q][:, n_per[q] : n_per[q] + hi - lo])
                n_per[q] += hi - lo
        ok = torch.equal(torch.cat(rebuilt, 1), ei.t()) and float(t) == 0.5 + world - 1
        torch.save(ok, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_edge_sharding_gloo(tmp_path):
    out = str(tmp_path / 'ok.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.load(out) is True


def test_shard_bounds_partition():
    from tgm_amd.dist import shard_bounds

    for n in (0, 1, 7, 200, 4096):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n and all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1


def _batch_worker(rank, world, port, out):
    """Batch-level sharding (DGDataLoader(batch_shard=(rank, world))): rank r walks batches r, r + world, ... of the unsharded
    schedule.  Host logic only: the slices, the schedule position handed to the negative sampler, the refusal of stateful hooks."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.dist import init_process_group
    from tgm_amd.hooks import HookManager, RecencyNeighborHook

    init_process_group('gloo')
    g = torch.Generator().manual_seed(0)
    E, bs = 103, 10
    ts = torch.sort(torch.randint(0, 50, (E,), generator=g)).values
    ei = torch.randint(0, 11, (E, 2), generator=g, dtype=torch.int32)
    dg = DGraph(DGData.from_raw(ts, ei, torch.rand(E, 2, generator=g)))
    loader = DGDataLoader(dg, batch_size=bs, batch_shard=(rank, world))
    mine = [(int(b._edge_lo), b.edge_src.clone()) for b in loader]
    assert len(loader) == len(mine) == len(range(rank * bs, E, bs * world))
    los = torch.full((8,), -1, dtype=torch.int64)
    los[: len(mine)] = torch.tensor([lo for lo, _ in mine])
    got = [torch.zeros(8, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(got, los)
    ok = True
    if rank == 0:
        inter = [int(got[j % world][j // world]) for j in range(-(-E // bs))]
        ok = inter == list(range(0, E, bs))  # interleaved, the ranks' batches ARE the unsharded schedule
    ok = ok and all(torch.equal(src, ei[lo : lo + bs, 0]) for lo, src in mine)
    # a sampler with per-batch state cannot skip the other ranks' batches
    hm = HookManager(keys=['k'])
    hm.register('k', RecencyNeighborHook(11, [2], ['edge_src', 'edge_dst'], ['edge_time', 'edge_time']))  # streaming rings
    with hm.activate('k'):
        try:
            next(iter(DGDataLoader(dg, batch_size=bs, hook_manager=hm, batch_shard=(rank, world))))
            ok = False
        except ValueError as e:
            ok = ok and 'carries state' in str(e)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        torch.save(bool(flag.item()), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_batch_sharding_gloo(tmp_path):
    out = str(tmp_path / 'ok.pt')
    mp.spawn(_batch_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.load(out) is True


def test_batch_shard_argument_checks():
    from tgm_amd import DGData, DGDataLoader, DGraph

    ts = torch.arange(20)
    dg = DGraph(DGData.from_raw(ts, torch.randint(0, 5, (20, 2), dtype=torch.int32), time_delta='s'))
    with pytest.raises(ValueError, match='rank'):
        DGDataLoader(dg, batch_size=4, batch_shard=(2, 2))
    with pytest.raises(ValueError, match='event-ordered'):
        DGDataLoader(dg, batch_size=4, batch_unit='s', batch_shard=(0, 2))
    assert [int(b._edge_lo) for b in DGDataLoader(dg, batch_size=4, batch_shard=(1, 3))] == [4, 16]
    assert [int(b._edge_lo) for b in DGDataLoader(dg, batch_size=4, batch_shard=(0, 3), drop_last=True)] == [0, 12]
    # 5 batches over 3 ranks: 2 / 2 / 1 without shard_even, 1 / 1 / 1 with it (every rank the same length: no collective is left waiting)
    assert [len(DGDataLoader(dg, batch_size=4, batch_shard=(r, 3))) for r in range(3)] == [2, 2, 1]
    even = [DGDataLoader(dg, batch_size=4, batch_shard=(r, 3), shard_even=True) for r in range(3)]
    assert [len(ld) for ld in even] == [1, 1, 1]
    assert [[int(b._edge_lo) for b in ld] for ld in even] == [[0], [4], [8]]
    assert [len(DGDataLoader(dg, batch_size=4, batch_shard=(r, 5), shard_even=True)) for r in range(5)] == [1] * 5

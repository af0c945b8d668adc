"""Two ranks (sharing the one test GPU, gloo collective staged through the host) run the TGN memory
update with the commit sharded across ranks + all-gather; every replica must equal the single-process
memory bit for bit.  On a multi-GPU node the same code path runs with backend nccl (RCCL)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, out):
    import torch.distributed as dist

    from tgm_amd.nn import IdentityMessage, LastAggregator, TGNMemory
    from tgm_amd.synth import make_stream

    if world > 1:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    st = make_stream('review', seed=5, num_edges=3000, n_src=500, n_dst=80)
    ts = st.ts[0] + torch.arange(st.num_edges) * 300
    N, D, M, T_ = st.num_nodes, 16, 32, 20
    torch.manual_seed(3)
    mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator()).to('cuda').train()
    for lo in range(0, st.num_edges, 256):
        hi = min(lo + 256, st.num_edges)
        mem.update_state(st.src[lo:hi].cuda(), st.dst[lo:hi].cuda(), ts[lo:hi].cuda(), st.edge_x[lo:hi].cuda())
    mem.eval()
    torch.save((mem.memory.cpu(), mem.last_update.cpu()), f'{out}.{world}.{rank}')
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_commit_equals_single_process(tmp_path):
    out = str(tmp_path / 'mem')
    _run(0, 1, 0, out)
    mp.spawn(_run, args=(2, _free_port(), out), nprocs=2, join=True)
    m1, l1 = torch.load(f'{out}.1.0')
    for r in (0, 1):
        m2, l2 = torch.load(f'{out}.2.{r}')
        assert torch.equal(m1, m2) and torch.equal(l1, l2), f'replica {r} diverged'
    assert m1.abs().sum() > 0

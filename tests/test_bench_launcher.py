"""`python bench.py --gpus N` started bare (the driver's command): bench.py re-launches itself as N ranks under
torch.distributed.run, rank 0 prints the one JSON line.  CPU part: the launcher builds the right command and passes the exit status
through.  GPU part: two ranks sharing the one test GPU (TGMX_SINGLE_DEVICE=1, gloo rendezvous) produce the line with the wiki
headline fields AND the `scale_comment` block (comment-shaped stream shrunk for the test) AND the `tgn_memory_allgather` block (the one
collective on the path: TGN memory commit rows sharded + all-gathered, replicas checksummed against a single-process run) -- the same
code runs one rank per GPU over RCCL on an 8-GPU node (`test_two_ranks_over_rccl`, skipped below two visible GPUs)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_multi_gpu_command_relaunches_under_torch_distributed_run(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '5', '--warmup', '2'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7  # the job's status is the command's status
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '5', '--warmup', '2']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_from_the_bare_command_print_one_line_with_scale_comment():
    env = dict(os.environ, TGMX_DIST_BACKEND='gloo', TGMX_SINGLE_DEVICE='1', TGMX_SCALE_COMMENT_EDGES='300000', TGMX_TGN_ALLGATHER_EDGES='60000')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '12', '--warmup', '4'],
                       env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 12 and out['scaling'] == 'weak'
    assert 'tgbl-wiki' in out['config']['workload'] and out['value'] > 0
    assert out['rccl']['ranks_seen'] == 2
    sc = out['scale_comment']
    assert 'error' not in sc, sc
    assert sc['rccl_ranks_seen'] == 2
    for mode in ('batch', 'weak'):
        blk = sc[mode]
        assert len(blk['per_rank_sampled_edges_per_s']) == 2 and min(blk['per_rank_sampled_edges_per_s']) > 0
        assert len(blk['per_rank_hop1_hbm_frac']) == 2 and 0 < min(blk['per_rank_hop1_hbm_frac']) <= 1.0
        # the aggregate is all ranks' slots over the slowest rank's barrier-to-barrier time: never more than the sum of the ranks' own rates
        assert 0 < blk['aggregate_sampled_edges_per_s'] <= sum(blk['per_rank_sampled_edges_per_s']) * 1.001
    assert sc['batch']['global_batch'] == 4096 and sc['weak']['global_batch'] == 8192
    _check_tgn_allgather(out['tgn_memory_allgather'], 'gloo', env)


def _check_tgn_allgather(ag, backend, env):
    """The block's contract + the replicas' checksum against the SAME steps in one process without any collective."""
    assert 'error' not in ag, ag
    assert ag['backend'] == backend and ag['ranks_seen'] == 2 and ag['replicas_identical'] is True
    assert ag['steps'] > 0 and ag['wall_us_per_step'] > 0 and min(ag['per_rank_step_us']) > 0
    for mean, lo, hi in ag['per_rank_allgather_us_mean_min_max']:
        assert 0 < lo <= mean <= hi
    # payload: ceil(rows / 2) records of (100 memory floats + one int64) per rank, both ranks' records received
    rows = ag['commit_rows_per_step']
    assert 2 <= rows <= 1024
    assert ag['bytes_sent_per_rank_per_step'] >= rows / 2 * 102 * 4 and ag['bytes_received_per_rank_per_step'] == 2 * ag['bytes_sent_per_rank_per_step']
    code = ('import json, argparse, torch, bench; '
            'print(json.dumps(bench.tgn_memory_allgather_block(argparse.Namespace(seed=1337), 0, 1, torch.device("cuda", 0))))')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    single = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert single['backend'] is None and single['per_rank_allgather_us_mean_min_max'] == [None]
    assert single['checksum'] == ag['checksum'] and single['checksum'][2] > 0, (single['checksum'], ag['checksum'])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_over_rccl():
    """The same bare command with one rank per GPU and the default backend (nccl = RCCL): runs wherever two GPUs are visible -- the first
    multi-GPU box exercises the RCCL all-gather without a code change."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip('needs two visible GPUs (backend nccl = RCCL over xGMI)')
    env = dict(os.environ, TGMX_SCALE_COMMENT_EDGES='300000', TGMX_TGN_ALLGATHER_EDGES='60000')
    for k in ('WORLD_SIZE', 'RANK', 'TGMX_DIST_BACKEND', 'TGMX_SINGLE_DEVICE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '12', '--warmup', '4'],
                       env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['rccl']['backend'] == 'nccl' and out['rccl']['ranks_seen'] == 2
    _check_tgn_allgather(out['tgn_memory_allgather'], 'nccl', env)

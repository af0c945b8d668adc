#!/usr/bin/env python
"""Headline benchmark: sampled-edges/sec + achieved HBM GB/s of the temporal
neighbor sampler (TGAT seeding, 2-hop, k=20) on a tgbl-wiki-shaped synthetic stream.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of the stream: slice the
resident COO store, draw negatives, sample 2 hops for [src | dst | neg] seeds
(hand-written HIP lookup kernels), append the batch to the per-node rings.  All
inputs are resident in HBM before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling -- the
global batch is N x 200 edges; every rank holds a replica of the stream, samples
for its own contiguous 200-edge slice and applies the whole batch's (tiny) ring
update.  No collective on the data path; barrier + device sync bracket the timed
region and the reported time is the max over ranks.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: "roofline" (dominant kernel = the hop-1 lookup/gather launch, timed with
HIP events on its launch stream inside the timed region) and "cpu_baseline" (the
reference algorithm restated for torch-CPU, oracle/ring_port.py, timed on this
box's host cores on a bounded sample of the same stream; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s achievable copy


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=2364)  # three passes over the 157 474-edge stream at bs=200 (788 batches each)
    p.add_argument('--warmup', type=int, default=100)
    p.add_argument('--workload', default='wiki', choices=['wiki', 'review', 'comment'])
    p.add_argument('--batch-size', type=int, default=None, help='edges per rank per step (default: 200 wiki, 512 review, 4096 comment)')
    p.add_argument('--num-nbrs', type=int, nargs='+', default=None)
    p.add_argument('--mode', default='ring', choices=['ring', 'csr'])
    p.add_argument('--cpu-batches', type=int, default=100, help='batches of the CPU-baseline sample (0 = skip); ~0.25 s each on the GPU box host')
    p.add_argument('--profile-every', type=int, default=32, help='bracket the dominant kernel with HIP events every n-th step')
    p.add_argument('--seed', type=int, default=1337)
    return p.parse_args()


DEFAULTS = {'wiki': (200, [20, 20]), 'review': (512, [10, 10]), 'comment': (4096, [20, 20])}


def build_pipeline(stream, rank, world, bs_rank, num_nbrs, mode, device):
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.dist import EdgeShardHook
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook

    data = DGData.from_raw(stream.ts, torch.stack([stream.src, stream.dst], 1), stream.edge_x, static_node_x=stream.node_x)
    dg = DGraph(data, device=device)
    global_bs = bs_rank * world
    hm = HookManager(keys=['bench'])
    lo_dst = int(stream.dst.min())
    if world > 1:
        hm.register('bench', EdgeShardHook(rank, world))
        keys, tkeys = ['shard_src', 'shard_dst', 'neg'], ['shard_time', 'shard_time', 'neg_time']
        hm.register('bench', RandomNegativeEdgeSamplerHook(lo_dst, stream.num_nodes, like='shard_dst', time_key='shard_time'))
    else:
        keys, tkeys = ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time']
        hm.register('bench', RandomNegativeEdgeSamplerHook(lo_dst, stream.num_nodes))
    hook = RecencyNeighborHook(
        stream.num_nodes, num_nbrs, keys, tkeys, mode=mode, validate='deferred', batch_size=global_bs if mode == 'csr' else None
    )
    hm.register('bench', hook)
    loader = DGDataLoader(dg, batch_size=global_bs, hook_manager=hm)
    return dg, hm, hook, loader


def cpu_baseline(stream, bs, num_nbrs, n_batches, seed):
    """Reference algorithm (torch-CPU tensor program, oracle/ring_port.py) on the host cores."""
    from oracle.ring_port import RingSamplerCPU

    src, dst, ts, x = stream.src.cpu(), stream.dst.cpu(), stream.ts.cpu(), None if stream.edge_x is None else stream.edge_x.cpu()
    D = 0 if x is None else x.shape[1]
    model = RingSamplerCPU(stream.num_nodes, num_nbrs, D)
    g = torch.Generator().manual_seed(seed)
    lo_dst = int(dst.min())
    E = src.numel()
    slots = 0
    t_total = 0.0
    done = 0
    for b in range(n_batches + 1):
        lo, hi = b * bs, min((b + 1) * bs, E)
        if lo >= E:
            break
        neg = torch.randint(lo_dst, stream.num_nodes, (hi - lo,), dtype=torch.int32, generator=g)
        seeds = torch.cat([src[lo:hi], dst[lo:hi], neg])
        times = torch.cat([ts[lo:hi]] * 3)
        t0 = time.perf_counter()
        hops = model.step(seeds, times, src[lo:hi], dst[lo:hi], ts[lo:hi], None if x is None else x[lo:hi])
        dt = time.perf_counter() - t0
        if b == 0:
            continue  # first batch warms the allocator
        t_total += dt
        slots += sum(h[2].numel() for h in hops)
        done += 1
    return dict(
        value=slots / t_total,
        unit='sampled-edges/s',
        cores=torch.get_num_threads(),
        kind='port',
        sample=f'batches 1..{done} of the same stream (bs={bs}, k={num_nbrs}), sampler stage only, {t_total:.1f} s',
        ms_per_step=1e3 * t_total / max(done, 1),
    )


def pmc_traffic(args, grid_slots):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    of this very command, separate runs; tools/gpu_round.sh writes profiles/pmc_hop1.json).  FETCH_SIZE is doubled: the
    gfx950 correction of MI355X_MICROARCH.md for wide coalesced reads.  None when no profile of this configuration exists."""
    path = os.path.join(ROOT, 'profiles', 'pmc_hop1.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        p = json.load(f)
    same = (p.get('workload'), p.get('mode'), p.get('batch_size'), p.get('num_nbrs')) == (
        args.workload, args.mode, args.batch_size or DEFAULTS[args.workload][0], args.num_nbrs or DEFAULTS[args.workload][1])
    if not same or p.get('slots_per_launch') != grid_slots:
        return None
    return {'bytes': 2 * 1024 * p['fetch_kb'] + 1024 * p['write_kb'], 'fetch_kb_x2': 2 * p['fetch_kb'], 'write_kb': p['write_kb'],
            'source': 'profiles/pmc_hop1.json (rocprofv3 --pmc, separate passes)'}


def rocprof_kernel_us(fused: bool):
    """Average duration of the dominant kernel in the committed rocprofv3 --kernel-trace --stats summary of this very
    command (profiles/r01_sampler_rocprof_summary.md; tools/gpu_round.sh writes it) -- the cross-check of the HIP-event
    figure (events bracket the launch from outside: ~2-3 us more than the kernel's own duration)."""
    path = os.path.join(ROOT, 'profiles', 'r01_sampler_rocprof_summary.md')
    if not os.path.exists(path):
        return None
    want = 'recency_lookup_fused01_kernel' if fused else 'recency_lookup_kernel'
    best = None
    for line in open(path):
        cells = [c.strip() for c in line.split('|')]
        if len(cells) > 5 and want in cells[1]:
            try:
                calls, avg = int(cells[2]), float(cells[4])
            except ValueError:
                continue
            if best is None or calls > best[0]:
                best = (calls, avg)
    return None if best is None else best[1]


def main():
    args = parse_args()
    from tgm_amd.dist import init_process_group
    from tgm_amd.synth import make_stream

    rank, world, local = init_process_group()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)'
    assert torch.cuda.is_available(), 'bench.py needs a ROCm device'
    if os.environ.get('TGMX_SINGLE_DEVICE'):  # functional check only: every rank on device 0 (with TGMX_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    bs_rank, num_nbrs = DEFAULTS[args.workload]
    bs_rank = args.batch_size or bs_rank
    num_nbrs = args.num_nbrs or num_nbrs
    # small shapes are generated on the host (bit-stable stream shared with the fixtures), big ones on the device
    gen_dev = 'cpu' if args.workload == 'wiki' else device
    stream = make_stream(args.workload, seed=args.seed, device=gen_dev)
    dg, hm, hook, loader = build_pipeline(stream, rank, world, bs_rank, num_nbrs, args.mode, device)
    D = stream.edge_dim
    n_batches = len(loader)
    S0 = 3 * bs_rank
    slots_per_step = 0
    S = S0
    for k in num_nbrs:
        slots_per_step += S * k
        S *= k
    last_hop = len(num_nbrs) - 1

    def run(n_steps, start):
        """n_steps consecutive batches, wrapping around the stream (epoch boundary = reset_state)."""
        it = start
        starts = loader._starts
        for _ in range(n_steps):
            if it == n_batches:
                hm.reset_state()
                it = 0
            loader(starts[it])
            it += 1
        return it

    with hm.activate('bench'):
        from tgm_amd._native import KernelTimer

        every = max(1, args.profile_every)
        # the warm-up runs with the same instrumentation as the timed steps (first-use costs of the counting ops land there)
        warm_every = max(1, min(every, args.warmup // 2))  # at least two instrumented warm-up steps
        hook.profile_hop, hook.profile_every, hook.profile_log = last_hop, warm_every, []
        hook.profile_pool = [KernelTimer() for _ in range(args.warmup // warm_every + 1)]
        pos = run(args.warmup, 0)
        hook.check()
        hook.profile_every, hook.profile_log = every, []
        # at most 48 timed launches: ~100 HIP events awaiting their timestamps is where the runtime starts to stall
        hook.profile_pool = [KernelTimer() for _ in range(min(48, args.steps // every + 1))]
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps, pos)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        hook.check()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (last hop's lookup + gather launch) -----------
    log = hook.profile_log
    hook.profile_hop = None
    ker_ms = [t.elapsed_ms() for t, *_ in log]
    avg_ms = sum(ker_ms) / max(len(ker_ms), 1)
    # the timed launch covers one hop, or hop 0 + hop 1 when tgmx_recency_step runs them as one launch
    shape = log[0][1] if log else []
    seeds_l = sum(seeds for seeds, _ in shape)
    total_slots = sum(seeds * k for seeds, k in shape)
    valid = sum(int(counts.sum().item()) for *_, counts in log) / max(len(log), 1)
    # algorithmic bytes per launch: every slot is written (id 4 + ts 8 + 4D),
    # valid slots also read their 16-byte record and 4D-byte feature row; 68 B of index traffic per seed
    algo_bytes = total_slots * (12 + 4 * D) + valid * (16 + 4 * D) + seeds_l * 68
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    fused = len(shape) > 1
    kernel_name = ('recency_lookup_fused01_kernel (hop 0 + hop 1 in one launch: ' if fused else f'recency_lookup_kernel (hop {last_hop}: ') + \
        ' + '.join(f'{seeds} seeds x k={k}' for seeds, k in shape) + ')'

    total_units = args.steps * slots_per_step * world
    out = {
        'metric': 'sampled-edges/sec (TGAT 2-hop k=20 recency sampler, tgbl-wiki synthetic)' if args.workload == 'wiki'
        else f'sampled-edges/sec (recency sampler, tgbl-{args.workload} synthetic)',
        'value': total_units / elapsed,
        'unit': 'sampled-edges/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'int32/int64 indices + f32 feature rows (copied, no arithmetic)',
        'data': 'synthetic',
        'config': {
            'workload': f'tgbl-{args.workload}-shaped synthetic stream: N={stream.num_nodes}, E={stream.num_edges}, D={D}; '
            f'seeds = src|dst|neg, num_nbrs={num_nbrs}, batch_size={bs_rank} edges per rank ({bs_rank * world} global), mode={args.mode}, '
            "seed validation on the device, read back once after the timed steps (validate='deferred')",
            'slots_per_step_per_rank': slots_per_step,
            'events_per_s': args.steps * bs_rank * world / elapsed,
            'parallelism': f'edge-batch sharding x{world}, replicated stream, no data-path collective',
        },
        'roofline': {
            'bound': 'hbm',
            'kernel': kernel_name,
            'achieved': achieved,
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            'traffic': pmc_traffic(args, total_slots),
            'avg_kernel_ms': avg_ms,
            'rocprof_avg_kernel_us': rocprof_kernel_us(fused) if (args.workload == 'wiki' and args.mode == 'ring' and world == 1) else None,
            'launches_timed': len(ker_ms),
            'algorithmic_bytes_per_launch': algo_bytes,
            'valid_slot_fraction': valid / max(total_slots, 1),
        },
    }
    if rank == 0:
        if world == 1 and args.cpu_batches > 0:
            out['cpu_baseline'] = cpu_baseline(stream, bs_rank, num_nbrs, args.cpu_batches, args.seed)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

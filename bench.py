#!/usr/bin/env python
"""Headline benchmark: sampled-edges/sec + achieved HBM GB/s of the temporal
neighbor sampler (TGAT seeding, 2-hop, k=20) on a tgbl-wiki-shaped synthetic stream.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of the stream: slice the
resident COO store, draw negatives, sample 2 hops for [src | dst | neg] seeds
(hand-written HIP lookup kernels), append the batch to the per-node rings.  All
inputs are resident in HBM before the timed region.

N > 1 (launched by torch.distributed.run, one rank per GPU): edge-batch sharding.
``--scaling weak`` (default): the global batch is N x 200 edges; ``--scaling strong``:
the global batch stays at the single-GPU size and rank r seeds from edges
[r * bs / N, (r + 1) * bs / N) of it.  Either way every rank
holds a replica of the stream and applies the whole batch's (tiny) ring update.
No collective on the data path; barrier + device sync bracket the timed region and
the reported time is the max over ranks.

Steady state: before the warm-up the stream is replayed UNTIMED up to ``--start-frac``
(default half) of its batches, so the timed steps run on filled rings (a fresh stream
would be measured on almost-empty rings: nearly all-pad output).  If the timed steps
reach the end of the stream the state is reset and the stream restarts (an epoch
boundary); the default step count stops before that.

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
    p.add_argument('--steps', type=int, default=None, help='timed batches (default: up to the end of the stream from --start-frac, at most 2000)')
    p.add_argument('--warmup', type=int, default=20)
    p.add_argument('--workload', default='wiki', choices=['wiki', 'review', 'comment'])
    p.add_argument('--batch-size', type=int, default=None, help='edges per rank per step (default: 200 wiki, 512 review, 4096 comment)')
    p.add_argument('--num-nbrs', type=int, nargs='+', default=None)
    p.add_argument('--mode', default='ring', choices=['ring', 'csr'])
    p.add_argument('--cpu-batches', type=int, default=30, help='batches of the CPU-baseline sample PER thread setting (0 = skip); ~0.1 s each: 30 x 4 settings = ~12 s of CPU work')
    p.add_argument('--scaling', default='weak', choices=['weak', 'strong', 'batch'],
                   help="N > 1: 'weak' = global batch N x bs, rank r seeds from its slice; 'strong' = global batch bs, split N ways; 'batch' = "
                   "the single-GPU schedule of bs-edge batches dealt round-robin to the ranks (DGDataLoader(batch_shard=), static index only: "
                   "the batches -- and therefore the sampled neighbours -- are exactly the single-GPU run's)")
    p.add_argument('--pool', type=int, default=1, help='DGDataLoader(output_pool=): ring of preallocated output sets, one native call per batch; '
                   '0 = hook-by-hook path with fresh tensors per batch (the reference-semantics default of the library)')
    p.add_argument('--validate', default='deferred', choices=['deferred', 'sync', 'off'], help="seed validation mode of the hook ('sync' = its default: a device->host read per batch)")
    p.add_argument('--start-frac', type=float, default=0.5, help='fraction of the stream replayed untimed before the warm-up (ring fill)')
    p.add_argument('--profile-every', type=int, default=32, help='bracket the dominant kernel with HIP events every n-th step')
    p.add_argument('--seed', type=int, default=1337)
    p.add_argument('--emulate-world', type=int, default=0, help='single process: run the step of rank --emulate-rank of a W-rank job (there is no data-path collective, so nothing is missing from it); modelling aid, n_gpus stays 1')
    p.add_argument('--emulate-rank', type=int, default=0)
    p.add_argument('--extras', default='auto', choices=['auto', 'on', 'off'],
                   help="the blocks next to the contract's fields -- the same launch with full feature writes / an output pool of two, the "
                   "HBM-bound comment-shaped lookup (roofline_hbm_bound), the TGAT aggregation block: 'auto' = at N = 1 on the wiki workload")
    p.add_argument('--scale-comment', default='on', choices=['on', 'off'],
                   help='N > 1: also measure the comment-shaped stream over the static index, batch-sharded and weak, in the same job (scale_comment)')
    p.add_argument('--tgn-allgather', default='on', choices=['on', 'off'],
                   help='N > 1: also time the TGN memory module with the commit sharded across the ranks and all-gathered (tgn_memory_allgather)')
    p.add_argument('--no-default-path', action='store_true', help="skip the second timed region (the same steps through DGDataLoader / RecencyNeighborHook with their DEFAULT arguments)")
    return p.parse_args()


def wake_host():
    """Between the synchronize in front of a timed region and the region's first step: keep the calling core busy for a few milliseconds.
    `torch.cuda.synchronize()` parks the thread while the device drains (after the untimed replay: several ms), and the core comes back from
    its idle state SLOW -- the next ~15-30 loader calls take 30-60 us of host time instead of 18-23 (tools/transient_after_sync.py:
    `headline 0` against `headline 3`, `default 0` against `default 3`).  A 20-step timed region would measure mostly that wake-up -- a
    property of the host's power management, once per blocking wait, not of the path.  Outside every timed region; TGMX_BENCH_SPIN_MS=0 turns
    it off (A/B)."""
    ms = float(os.environ.get('TGMX_BENCH_SPIN_MS', '3'))
    t = time.perf_counter()
    while time.perf_counter() - t < ms * 1e-3:
        pass


DEFAULTS = {'wiki': (200, [20, 20]), 'review': (512, [10, 10]), 'comment': (4096, [20, 20])}


def build_pipeline(stream, rank, world, global_bs, num_nbrs, mode, device, pool=0, validate='deferred', edge_features='dense', batch_shard=False):
    from tgm_amd import DGData, DGDataLoader, DGraph
    from tgm_amd.dist import EdgeShardHook
    from tgm_amd.hooks import HookManager, RandomNegativeEdgeSamplerHook, RecencyNeighborHook

    data = DGData.from_raw(stream.ts, torch.stack([stream.src, stream.dst], 1), stream.edge_x, static_node_x=stream.node_x)
    dg = DGraph(data, device=device)
    hm = HookManager(keys=['bench'])
    lo_dst = int(stream.dst.min())
    if world > 1 and not batch_shard:
        hm.register('bench', EdgeShardHook(rank, world))
        keys, tkeys = ['shard_src', 'shard_dst', 'neg'], ['shard_time', 'shard_time', 'neg_time']
        hm.register('bench', RandomNegativeEdgeSamplerHook(lo_dst, stream.num_nodes, like='shard_dst', time_key='shard_time'))
    else:
        keys, tkeys = ['edge_src', 'edge_dst', 'neg'], ['edge_time', 'edge_time', 'neg_time']
        hm.register('bench', RandomNegativeEdgeSamplerHook(lo_dst, stream.num_nodes))
    kw = {} if validate is None else {'validate': validate}  # None: the hook's own default ('sync', the reference's raise-per-call)
    if edge_features != 'dense':
        kw['edge_features'] = edge_features
    hook = RecencyNeighborHook(stream.num_nodes, num_nbrs, keys, tkeys, mode=mode, batch_size=global_bs if mode == 'csr' else None, **kw)
    hm.register('bench', hook)
    kw_l = {'batch_shard': (rank, world)} if (batch_shard and world > 1) else {}
    if pool is None:  # the loader's own default: what an unmodified TGM script gets
        loader = DGDataLoader(dg, batch_size=global_bs, hook_manager=hm, **kw_l)
    else:
        loader = DGDataLoader(dg, batch_size=global_bs, hook_manager=hm, output_pool=pool, **kw_l)
    return dg, hm, hook, loader


def cpu_baseline(stream, bs, num_nbrs, n_batches, seed, first_batch):
    """Reference algorithm (torch-CPU tensor program, oracle/ring_port.py) on the host cores: the same region of the same
    stream as the timed GPU steps (the rings are brought there with update-only calls, untimed), best of a few thread counts
    (small tensor ops oversubscribe a 128-core host: all cores is rarely the fastest)."""
    from oracle.ring_port import RingSamplerCPU

    src, dst, ts, x = stream.src.cpu(), stream.dst.cpu(), stream.ts.cpu(), None if stream.edge_x is None else stream.edge_x.cpu()
    D = 0 if x is None else x.shape[1]
    model = RingSamplerCPU(stream.num_nodes, num_nbrs, D)
    g = torch.Generator().manual_seed(seed)
    lo_dst = int(dst.min())
    E = src.numel()
    restore_threads = torch.get_num_threads()
    all_threads = os.cpu_count() or restore_threads  # (main() pins the process to 16 torch threads; this leg tries up to 64)
    b = first_batch
    results = {}
    t_all = 0.0
    try:
        torch.set_num_threads(min(16, all_threads))  # (the untimed fill too: small tensor ops on every core of a shared host can take minutes)
        for bb in range(max(0, first_batch - 400), first_batch):  # fill the rings: update only (the 400 batches before the sample)
            lo, hi = bb * bs, min((bb + 1) * bs, E)
            model.update(src[lo:hi], dst[lo:hi], ts[lo:hi], None if x is None else x[lo:hi])
        # (up to 64 threads: every setting above 32 has been slower on these hosts, and "all cores" of a SHARED 256-core host is where a
        # run of this leg once took minutes -- the OpenMP barriers of small tensor ops under oversubscription; each setting is also
        # bounded in wall time: the leg is a reported baseline, it must not decide how long the bench runs)
        budget_s = float(os.environ.get('TGMX_CPU_BASELINE_SECONDS', 6.0))
        for threads in sorted({t for t in (8, 16, 32, min(64, all_threads)) if t <= all_threads}):
            torch.set_num_threads(threads)
            slots, t_total, done = 0, 0.0, 0
            t_set = time.perf_counter()
            for i in range(n_batches + 1):
                if done >= 2 and time.perf_counter() - t_set > budget_s:
                    break
                lo, hi = b * bs, min((b + 1) * bs, E)
                if lo >= E:
                    break
                b += 1
                neg = torch.randint(lo_dst, stream.num_nodes, (hi - lo,), dtype=torch.int32, generator=g)
                seeds = torch.cat([src[lo:hi], dst[lo:hi], neg])
                times = torch.cat([ts[lo:hi]] * 3)
                t0 = time.perf_counter()
                hops = model.step(seeds, times, src[lo:hi], dst[lo:hi], ts[lo:hi], None if x is None else x[lo:hi])
                dt = time.perf_counter() - t0
                if i == 0:
                    continue  # first batch at this thread count warms the pool / allocator
                t_total += dt
                slots += sum(h[2].numel() for h in hops)
                done += 1
            if done:
                results[threads] = (slots / t_total, 1e3 * t_total / done, done)
                t_all += t_total
    finally:
        torch.set_num_threads(restore_threads)
    best = max(results, key=lambda t: results[t][0])
    return dict(
        value=results[best][0],
        unit='sampled-edges/s',
        cores=best,
        kind='port',
        sample=f'{results[best][2]} batches per thread setting from batch {first_batch} of the same stream (bs={bs}, k={num_nbrs}, rings filled by '
        f'the preceding batches), sampler stage only, {t_all:.1f} s of timed CPU work in total; best of threads={sorted(results)}',
        ms_per_step=results[best][1],
        by_threads={str(t): {'sampled_edges_per_s': v[0], 'ms_per_step': v[1]} for t, v in results.items()},
        host_cores=os.cpu_count(),
    )


def profile_key(args, bs, num_nbrs, steps, world=1):
    """What a committed profile must have been taken with to describe THIS run's timed launches."""
    return (f'{args.workload}|{args.mode}|bs{bs}|k{"x".join(map(str, num_nbrs))}|steps{steps}|warmup{args.warmup}|pool{args.pool}|{args.validate}|start{args.start_frac}'
            + (f'|world{world}|{args.scaling}' if world > 1 else ''))


def committed_profile(key):
    """(pmc json, its path) of the newest committed profile taken with exactly these arguments (tools/gpu_profile_r6.sh), or (None, None)."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_*_pmc.json')), reverse=True):  # the newest round's first
        try:
            with open(path) as f:
                p = json.load(f)
        except Exception:
            continue
        if p.get('profile_key') == key and p.get('fetch_kb') is not None and p.get('write_kb') is not None:
            return p, path
    return None, None


def kernel_src_sha():
    """Identity of the sampler's kernel sources in THIS tree (there is no .git on the GPU box): a committed counter profile describes the
    running kernels only if it was taken from the same sources (tools/gpu_profile_r6.sh records this value)."""
    import hashlib

    h = hashlib.sha256()
    for name in ('recency.hip', 'pipeline.hip', 'common.h'):
        with open(os.path.join(ROOT, 'tgm_amd', 'csrc', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(key, note=None):
    """HBM-side bytes per TIMED launch of the dominant kernel from the committed PMC passes of this very command (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, separate runs, averaged over the timed launches only: tools/gpu_profile_r6.sh).  FETCH_SIZE is doubled:
    the gfx950 correction of MI355X_MICROARCH.md for wide coalesced reads.  None when no profile of these arguments exists, or when
    the newest one was taken from other kernel sources than this tree's (`note` then says which file was passed over and why)."""
    p, path = committed_profile(key)
    if p is None:
        return None
    sha = kernel_src_sha()
    if p.get('kernel_src_sha') != sha:
        if note is not None:
            note['traffic_omitted'] = (f'{os.path.relpath(path, ROOT)} was taken from kernel sources {p.get("kernel_src_sha") or "(unrecorded: a round-4 profile)"}, '
                                       f'this tree is {sha}: its counters do not describe the launches timed here')
        return None
    return {'profile_file': os.path.relpath(path, ROOT), 'profile_kernel_src_sha': p.get('kernel_src_sha'), 'profile_taken': p.get('taken'),
            'running_kernel_src_sha': sha,'bytes': 2 * 1024 * p['fetch_kb'] + 1024 * p['write_kb'], 'fetch_kb_x2': 2 * p['fetch_kb'], 'write_kb': p['write_kb'],
            'dispatches_counted': p.get('dispatches_counted'),
            'source': f'{os.path.relpath(path, ROOT)}: rocprofv3 --pmc in separate passes of `python bench.py {p.get("bench_args")}`, {p.get("which")} '
                      '(a separate run of the same command, not this process)'}


def rocprof_kernel_us(key):
    """Average duration of the dominant kernel over the TIMED launches in the committed rocprofv3 --kernel-trace --stats summary taken
    with exactly these arguments (profiles/r0N_<tag>_rocprof_summary.md next to the matching r0N_<tag>_pmc.json).  Cross-check of
    the figure measured live (the dispatch's own begin / end timestamps through hipExtLaunchKernelGGL); a separate run."""
    p, path = committed_profile(key)
    if p is None:
        return None
    md = path.replace('_pmc.json', '_rocprof_summary.md')
    if not os.path.exists(md):
        return None
    for line in open(md):
        cells = [c.strip() for c in line.split('|')]
        if len(cells) > 3 and cells[1].startswith('timed region'):
            try:
                return {'avg_us': float(cells[2].split()[1]), 'source': f'{os.path.relpath(md, ROOT)} (rocprofv3 --kernel-trace of the same command, a separate run: {cells[1]})'}
            except (ValueError, IndexError):
                return None
    return None


def slots_of_shape(edges, num_nbrs):
    total, S = 0, 3 * edges
    for k in num_nbrs:
        total += S * k
        S *= k
    return total


def launch_stats(log, D):
    """Per-launch averages over the timed launches of the dominant kernel (hook.profile_log): duration, shape, and the VALID-AWARE
    algorithmic bytes: every slot's id 4 + ts 8 is written, a feature row (4D) for every slot that has to
    change; only valid slots read their 16-byte record and 4D-byte feature row (pads are zeros written without reading anything);
    68 B of index traffic per seed (+ 8 B of valid-count read / write per row with delta writes).  Also SURVEY 8(d)'s figure taken
    LITERALLY -- (28 + 8D) per slot + 68 per seed, pads charged a record + feature read they never make."""
    ker_ms = [e[0].elapsed_ms() for e in log]
    avg_ms = sum(ker_ms) / len(ker_ms)
    # the timed launch covers one hop, or hop 0 + hop 1 when tgmx_recency_step runs them as one launch
    seeds_l = sum(sum(seeds for seeds, _ in e[1]) for e in log) / len(log)
    total_slots = sum(sum(seeds * k for seeds, k in e[1]) for e in log) / len(log)
    valid = sum(int(e[2].sum().item()) for e in log) / len(log)
    delta = all(e[3] is not None for e in log)
    # feature rows written per launch: every slot, or -- pooled outputs with delta writes (tgmx_recency_step_t.out_valid) -- the
    # slots from the first one that changes on, max(valid before, valid now) per row, measured on the timed launches
    feat_slots = (sum(int(e[3].sum().item()) for e in log) / len(log)) if delta else total_slots
    algo = total_slots * 12 + feat_slots * 4 * D + valid * (16 + 4 * D) + seeds_l * (68 + (8 if delta else 0))
    full = total_slots * (12 + 4 * D) + valid * (16 + 4 * D) + seeds_l * 68
    literal = total_slots * (28 + 8 * D) + seeds_l * 68
    return dict(avg_ms=avg_ms, ker_ms=ker_ms, seeds=seeds_l, slots=total_slots, valid=valid, delta=delta, feat_slots=feat_slots, algo_bytes=algo,
                full_write_bytes=full, literal_8d_bytes=literal, shape=log[0][1])


def probe_variant(stream, bs, num_nbrs, mode, device, first_timed, n_steps, pool, env=None, rank=0, world=1, batch_shard=False, barrier=False):
    """The dominant launch of the SAME timed batches under another configuration (full feature writes, an output pool of two, another
    stream shape, a rank's share of a multi-rank job ...): a fresh pipeline replays the stream up to the timed region untimed, then
    `n_steps` steps, ~8 of them timed.  `bs` is the GLOBAL batch; with world > 1 this rank takes its slice of every batch, or
    (batch_shard) every world-th batch of the schedule, and `barrier` brackets the timed region with the process group's barrier."""
    from tgm_amd._native import KernelTimer

    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        dg, hm, hook, loader = build_pipeline(stream, rank, world, bs, num_nbrs, mode, device, pool=pool, validate='deferred', batch_shard=batch_shard)
        starts = loader._starts
        first_timed = max(0, min(first_timed, len(starts) - n_steps - 1))
        with hm.activate('bench'):
            warm = min(20, first_timed)
            for i in range(first_timed - warm):
                loader(starts[i])
            every = max(1, n_steps // 8)
            hook.profile_hop, hook.profile_every, hook.profile_log = len(num_nbrs) - 1, max(1, warm // 2), []
            hook.profile_pool = [KernelTimer() for _ in range(4)]
            for i in range(first_timed - warm, first_timed):
                loader(starts[i])
            hook.check()
            hook.profile_every, hook.profile_log, hook._calls = every, [], 0
            hook.profile_pool = [KernelTimer() for _ in range(n_steps // every + 1)]
            torch.cuda.synchronize()
            if barrier:
                torch.distributed.barrier()
            wake_host()
            t0 = time.perf_counter()
            for i in range(first_timed, first_timed + n_steps):
                loader(starts[i])
            torch.cuda.synchronize()
            wall_own = time.perf_counter() - t0
            if barrier:
                torch.distributed.barrier()
            wall = time.perf_counter() - t0
            hook.check()
            st = launch_stats(hook.profile_log, stream.edge_dim)
            hook.profile_hop = None
        st['us_per_step'] = 1e6 * wall / n_steps
        st['us_per_step_own'] = 1e6 * wall_own / n_steps
        st['first_timed'] = first_timed
        st['edges_timed'] = sum(min(bs, stream.num_edges - starts[i]) for i in range(first_timed, first_timed + n_steps))
        return st
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def scale_comment_block(args, rank, world, device, n_steps=48):
    """N > 1: north_star's scaling shape measured in the same job -- comment-shaped stream (N = 1 M, E = 44 M, D = 16; replicated in
    every GPU's HBM), bs = 4096, k = [20, 20], static index (the multi-GPU mode: no per-batch state, no data-path collective).
    'batch': the single-GPU schedule of 4096-edge batches dealt round-robin to the ranks (DGDataLoader(batch_shard=)) -- the sampled
    neighbours are the single-GPU run's; 'weak': a global batch of N x 4096 edges, rank r seeding from its slice (EdgeShardHook).
    Both: every rank runs `n_steps` steps between two barriers; aggregate = all ranks' sampled slots / the slowest rank's wall time."""
    import torch.distributed as dist

    from tgm_amd.synth import make_stream

    t_c = time.perf_counter()
    edges = int(os.environ.get('TGMX_SCALE_COMMENT_EDGES', 0)) or None  # tests shrink the stream; the bench does not
    cs = make_stream('comment', seed=args.seed, device=device, num_edges=edges)
    cbs, cnb = DEFAULTS['comment']
    D = cs.edge_dim
    out = {'workload': f'tgbl-comment-shaped synthetic stream: N={cs.num_nodes}, E={cs.num_edges}, D={D}; bs={cbs}, k={cnb}, static index '
                       f'(mode=csr), pool of one, delta writes; {world} ranks, replicated stream, no data-path collective; {n_steps} timed steps per rank '
                       'from the middle of the stream, barrier on both sides'}
    for scaling in ('batch', 'weak'):
        by_batch = scaling == 'batch'
        gbs = cbs if by_batch else cbs * world
        n_sched = (cs.num_edges + gbs - 1) // gbs
        n_own = len(range(rank, n_sched, world)) if by_batch else n_sched
        steps_r = min(n_steps, max(1, n_own // 2 - 2))  # the same on every rank (+-1 batch of n_own never reaches it at these sizes)
        st = probe_variant(cs, gbs, cnb, 'csr', device, n_own // 2, steps_r, pool=1, rank=rank, world=world, batch_shard=by_batch, barrier=True)
        # this rank's seed edges over its timed steps (full batches in the middle of the stream): whole batches, or its slice of each
        own_edges = st['edges_timed'] if by_batch else steps_r * ((gbs * (rank + 1)) // world - (gbs * rank) // world)
        mine = {'rank': rank, 'slots': slots_of_shape(own_edges, cnb),
                'us_per_step_own': st['us_per_step_own'], 'wall_us_per_step': st['us_per_step'], 'steps': steps_r,
                'hop1_kernel_ms': st['avg_ms'], 'hop1_algorithmic_bytes': st['algo_bytes'],
                'hop1_hbm_frac': st['algo_bytes'] / (st['avg_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        wall_s = max(r['wall_us_per_step'] * r['steps'] for r in ranks) * 1e-6
        out[scaling] = {
            'global_batch': gbs, 'aggregate_sampled_edges_per_s': sum(r['slots'] for r in ranks) / wall_s,
            'per_rank_sampled_edges_per_s': [r['slots'] / (r['us_per_step_own'] * r['steps'] * 1e-6) for r in ranks],
            'per_rank_us_per_step': [r['us_per_step_own'] for r in ranks],
            'per_rank_hop1_hbm_frac': [r['hop1_hbm_frac'] for r in ranks], 'per_rank_hop1_kernel_ms': [r['hop1_kernel_ms'] for r in ranks],
            'kernel': 'lookup_tile_coop_kernel (hop 1: %d seeds x k=%d per rank)' % (st['shape'][-1][0], st['shape'][-1][1]),
        }
    del cs
    torch.cuda.empty_cache()
    out['seconds_spent'] = time.perf_counter() - t_c
    return out


def tgn_memory_allgather_block(args, rank, world, device, n_steps=100, n_warm=30):
    """The one collective north_star names, timed: BASELINE cfg 3's memory module (review-shaped stream, bs = 512 global batch,
    TGNMemory(Last aggregation, GRU, memory 100, time 100), train mode) with the batch's edges sharded over the ranks.  Per step, on
    every rank: ``memory(n_id)`` for the unique endpoints of the rank's SLICE of the batch (the rows its share of the embedding
    reads: messages -> aggregation -> GRU, nothing written), then ``update_state`` of the whole batch with the commit sharded --
    each rank evaluates rows / N of the commit rows from its replica, ONE ``all_gather_into_tensor`` of fixed-size (memory row,
    last_update) records (RCCL over xGMI under backend nccl) gives every replica all rows (tgm_amd/nn/tgn.py ``_updated_sharded``;
    semantics tgm/nn/encoder/tgn.py:165-229).  The all-gather is bracketed by HIP events on the stream it is issued from; the
    replicas are compared by an exact (integer) checksum of memory and last_update at the end.  ``world == 1`` (the tests'
    single-process leg): the same steps unsharded, no collective -- its checksum is what the replicas must equal."""
    from tgm_amd.dist import shard_bounds
    from tgm_amd.nn import IdentityMessage, LastAggregator, TGNMemory
    from tgm_amd.synth import make_stream

    dist = torch.distributed
    t_c = time.perf_counter()
    edges = int(os.environ.get('TGMX_TGN_ALLGATHER_EDGES', 0)) or 400_000  # a 400 k-edge prefix-shaped stream: the steps touch a few hundred batches
    rs = make_stream('review', seed=args.seed, device=device, num_edges=edges)
    gbs, M, T_ = DEFAULTS['review'][0], 100, 100
    D, N = rs.edge_dim, rs.num_nodes
    torch.manual_seed(args.seed)  # every replica starts from the same parameters
    mem = TGNMemory(N, D, M, T_, IdentityMessage(D, M, T_), LastAggregator()).to(device).train()
    mem.shard_commits = True
    n_batches = rs.num_edges // gbs
    n_warm = min(n_warm, max(1, n_batches // 4))
    n_steps = min(n_steps, n_batches - n_warm)
    # inputs resident before the timed region: per batch the unique endpoints of this rank's slice of the edges
    share = []
    for b in range(n_warm + n_steps):
        lo, hi = shard_bounds(gbs, rank, world)
        e = slice(b * gbs + lo, b * gbs + hi)
        share.append(torch.unique(torch.cat([rs.src[e], rs.dst[e]])))

    def step(b):
        e = slice(b * gbs, (b + 1) * gbs)
        with torch.no_grad():
            mem(share[b])
            mem.update_state(rs.src[e], rs.dst[e], rs.ts[e], rs.edge_x[e])

    for b in range(n_warm):
        step(b)
    mem.allgather_log = log = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(n_warm, n_warm + n_steps):
        step(b)
    torch.cuda.synchronize()
    own_s = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    wall_s = time.perf_counter() - t0
    mem.allgather_log = None
    ag_us = [1e3 * a.elapsed_time(b_) for a, b_, *_ in log]
    # exact checksums: the float table summed as its int32 bit patterns (any differing bit moves the sum), int64 last_update as is
    checksum = [int(mem.memory.view(torch.int32).to(torch.int64).sum().item()), int(mem.last_update.sum().item()),
                int((mem.last_update != 0).sum().item())]
    mine = {'rank': rank, 'step_us': 1e6 * own_s / n_steps, 'checksum': checksum,
            'allgather_us': [sum(ag_us) / len(ag_us), min(ag_us), max(ag_us)] if ag_us else None,
            'commit_rows_per_step': sum(r for _, _, r, _, _ in log) / len(log) if log else None,
            'bytes_sent_per_step': sum(bs_ for _, _, _, bs_, _ in log) / len(log) if log else 0,
            'bytes_received_per_step': sum(br for _, _, _, _, br in log) / len(log) if log else 0}
    ranks = [mine]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        wall_s = max(wall_s, max(r['step_us'] for r in ranks) * 1e-6 * n_steps)
    del rs, mem
    torch.cuda.empty_cache()
    return {
        'workload': f'BASELINE cfg 3 memory module: tgbl-review-shaped synthetic stream (N={N}, first {edges} edges, D={D}), global batch {gbs} edges, '
                    f'TGNMemory(Last, GRU, memory {M}, time {T_}) in train mode, {world} rank(s); per step memory(n_id) over the unique endpoints of the '
                    f"rank's {gbs // world}-edge slice + update_state(whole batch) with the commit rows sharded and all-gathered; {n_steps} timed steps after "
                    f'{n_warm}, barrier on both sides',
        'backend': dist.get_backend() if world > 1 else None,
        'ranks_seen': len(ranks),
        'steps': n_steps,
        'wall_us_per_step': 1e6 * wall_s / n_steps,
        'events_per_s': gbs * n_steps / wall_s,
        'per_rank_step_us': [r['step_us'] for r in ranks],
        'per_rank_allgather_us_mean_min_max': [r['allgather_us'] for r in ranks],
        'allgather_what': 'one all_gather_into_tensor per step of ceil(rows / N) x (memory_dim + 2) float32 words per rank ((memory row, int64 last_update) '
                          'records), HIP events on the issuing stream; under gloo (functional tests) the figure includes the staging copies through the host',
        'commit_rows_per_step': mine['commit_rows_per_step'],
        'bytes_sent_per_rank_per_step': mine['bytes_sent_per_step'],
        'bytes_received_per_rank_per_step': mine['bytes_received_per_step'],
        'checksum': checksum,
        'replicas_identical': all(r['checksum'] == checksum for r in ranks),
        'seconds_spent': time.perf_counter() - t_c,
    }


def tgat_gflop_folded(S0, num_nbrs, node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, heads=2):
    """Multiply-adds x 2 of the FOLDED inference forward at the reference example's widths: per row of layer j
    d H C (folded queries) + 2 k H C (scores + weighted mean) + O C (W_V, all heads) + O^2 (W_O) + emb (O + d0) (fc1) + emb^2 (fc2)."""
    L = len(num_nbrs)
    rows = [S0]
    for k in num_nbrs:
        rows.append(rows[-1] * k)
    total = 0.0
    for j in range(1, L + 1):
        d = node_dim if j == 1 else embed_dim
        C = d + edge_dim + time_dim
        O = d + time_dim
        O += (-O) % heads
        R = sum(rows[: L - j + 1])
        k = num_nbrs[j - 1]
        total += R * (d * heads * C + 2 * k * heads * C + O * C + O * O + embed_dim * (O + node_dim) + embed_dim * embed_dim)
    return 2.0 * total / 1e9


def committed_mfma():
    """MFMA-busy fractions of the forward from the newest committed counter pass (profiles/r0N_tgat_mfma_pmc.json: rocprofv3 --pmc
    SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/bench_tgat.py, a separate run), or Nones.  The file records the hash of
    csrc/tgat.hip it was taken from; a pass over other sources is named as such in the source string."""
    import glob
    import hashlib

    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_tgat_mfma_pmc.json')), reverse=True)
    if not paths:
        return None, None, None
    path = paths[0]
    try:
        with open(path) as f:
            p = json.load(f)
        with open(os.path.join(ROOT, 'tgm_amd', 'csrc', 'tgat.hip'), 'rb') as f:
            sha = hashlib.sha256(f.read()).hexdigest()[:16]
        rel = os.path.relpath(path, ROOT)
        if p.get('tgat_src_sha') != sha:
            rel += f' [taken from csrc/tgat.hip {p.get("tgat_src_sha") or "(unrecorded)"}; this tree is {sha}: an OLDER kernel]'
        return p.get('mfma_busy_frac_forward'), p.get('mfma_busy_frac_tail_kernel'), rel
    except Exception:
        return None, None, None


def aggregation_block(stream, bs, num_nbrs, device, n_batches, first):
    """The other half of BASELINE.json's path: TGAT eval forward (reference example widths: node 1 / edge 172 / time 100 / embed 172,
    2 heads, one layer per hop) on the sampler's outputs at the headline batch shape -- RecencyNeighborHook(edge_features='by_id')
    feeding tgm_amd.nn.TGAT, random-init weights.  forward_us: median of 5 x 20 back-to-back forwards on one batch (HIP events);
    sampler_plus_forward_us: wall clock per batch over `n_batches` batches of loader + forward, synchronized at the end."""
    from tgm_amd.nn import TGAT

    dg, hm, hook, loader = build_pipeline(stream, 0, 1, bs, num_nbrs, 'ring', device, pool=1, edge_features='by_id')
    torch.manual_seed(0)
    d0 = int(stream.node_x.shape[1])
    enc = TGAT(node_dim=d0, edge_dim=stream.edge_dim, time_dim=100, embed_dim=172, num_layers=len(num_nbrs)).to(device).eval()
    starts = loader._starts
    node_x = dg.static_node_x
    first = max(1, min(first, len(starts) - n_batches - 1))
    with hm.activate('bench'), torch.no_grad():
        for i in range(first):
            b = loader(starts[i])
        for _ in range(10):  # the first forwards pay one-off costs (weight fold, code-object load)
            enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
        torch.cuda.synchronize()
        reps = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
            e1.record()
            torch.cuda.synchronize()
            reps.append(e0.elapsed_time(e1) / 20 * 1000)
        S0 = int(b.seed_nids[0].numel())
        t0 = time.perf_counter()
        for i in range(first, first + n_batches):
            b = loader(starts[i])
            enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        hook.check()
    fwd = sorted(reps)[len(reps) // 2]
    gflop = tgat_gflop_folded(S0, num_nbrs, d0, stream.edge_dim)
    busy_fwd, busy_tail, src = committed_mfma()
    return {
        'what': f"TGAT eval forward on the sampler's outputs (edge features by id), example widths node {d0} / edge {stream.edge_dim} / time 100 / embed 172, "
                f'2 heads, {len(num_nbrs)} layers, {S0} seeds, k={num_nbrs}; random-init weights; f32 (exact-fp32 MFMA for the dense contractions)',
        'forward_us': fwd, 'forward_us_min_max': [min(reps), max(reps)],
        'forward_repeats': '5 x 20 back-to-back forwards on one batch, HIP events on the launch stream; median',
        'sampler_plus_forward_us': 1e6 * (t1 - t0) / n_batches, 'batches': n_batches,
        'gflop_folded': gflop, 'tflops_folded': gflop / fwd * 1e3, 'mfma_peak_tflops_fp32': 157.3,
        'mfma_busy_frac': busy_fwd, 'mfma_busy_frac_tail_kernel': busy_tail,
        'source': 'forward_us / sampler_plus_forward_us: this process; mfma_busy_frac: '
                  + (f'{src} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/bench_tgat.py, a separate run: busy / (32 x cycles) '
                     'over the whole forward, and over the one-kernel tail of layer 1 alone)' if src else 'no committed counter pass'),
    }


def self_launch(n):
    """`python bench.py --gpus N ...` started bare (no WORLD_SIZE in the environment): re-run this very command line as N ranks of ONE
    node under torch.distributed.run (one process per GPU, RCCL; rendezvous on 127.0.0.1 at a free port), pass rank 0's JSON line through
    and exit with the job's status.  Under the driver's own `python -m torch.distributed.run ... bench.py --gpus N` WORLD_SIZE is set
    and this is never reached."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and not args.emulate_world:
        raise SystemExit(self_launch(args.gpus))
    from tgm_amd.dist import init_process_group
    from tgm_amd.synth import make_stream

    # host-side tensor work of this process (stream generation, DGData's checks) on 16 threads: torch's default is every core, and on a SHARED
    # 256-core host the OpenMP barriers of small ops under oversubscription made single runs take a minute longer than others
    torch.set_num_threads(min(16, torch.get_num_threads()))
    rank, world, local = init_process_group()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    real_world = world
    if args.emulate_world:
        assert world == 1, '--emulate-world is a single-process modelling aid'
        rank, world = args.emulate_rank, args.emulate_world
    assert torch.cuda.is_available(), 'bench.py needs a ROCm device'
    if os.environ.get('TGMX_SINGLE_DEVICE'):  # functional check only: every rank on device 0 (with TGMX_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    bs, num_nbrs = DEFAULTS[args.workload]
    bs = args.batch_size or bs
    num_nbrs = args.num_nbrs or num_nbrs
    # weak: every rank seeds from bs edges of a (world x bs)-edge global batch; strong: the global batch stays bs edges
    by_batch = args.scaling == 'batch'
    if by_batch and args.mode != 'csr':
        raise SystemExit("bench.py: --scaling batch needs --mode csr (whole batches are independent units only over the static index; "
                         'streaming rings carry state from batch to batch)')
    global_bs = bs * world if args.scaling == 'weak' else bs
    lo_r, hi_r = (global_bs * rank) // world, (global_bs * (rank + 1)) // world
    bs_rank = global_bs if by_batch else hi_r - lo_r  # this rank's seed edges per full batch
    # small shapes are generated on the host (bit-stable stream shared with the fixtures), big ones on the device
    gen_dev = 'cpu' if args.workload == 'wiki' else device
    stream = make_stream(args.workload, seed=args.seed, device=gen_dev)
    dg, hm, hook, loader = build_pipeline(stream, rank, world, global_bs, num_nbrs, args.mode, device, pool=args.pool, validate=args.validate,
                                          batch_shard=by_batch)
    D = stream.edge_dim
    n_batches = len(loader)
    last_hop = len(num_nbrs) - 1

    def slots_of(edges):
        total, S = 0, 3 * edges
        for k in num_nbrs:
            total += S * k
            S *= k
        return total

    # ---- schedule: [0, first) untimed ring fill, then the warm-up, then the timed steps -----------------
    first_timed = min(max(int(args.start_frac * n_batches), args.warmup), n_batches - 1)
    steps = args.steps if args.steps is not None else max(1, min(2000, n_batches - 1 - first_timed))  # stop before the ragged last batch
    fill = first_timed - args.warmup
    starts = loader._starts
    state = {'it': 0, 'edges': 0}

    def run(n_steps, count=False):
        """n_steps consecutive batches, wrapping around the stream (epoch boundary = reset_state)."""
        it = state['it']
        edges = 0
        for _ in range(n_steps):
            if it == n_batches:
                hm.reset_state()
                it = 0
            loader(starts[it])
            if count:
                e_lo = starts[it]
                edges += min(global_bs, stream.num_edges - e_lo)
            it += 1
        state['it'] = it
        return edges

    with hm.activate('bench'):
        from tgm_amd._native import KernelTimer

        run(fill)  # untimed: brings the rings to their state at `first_timed - warmup`
        every = max(1, min(args.profile_every, steps // 4))  # at least four timed launches, however short the run
        # the warm-up runs with the same instrumentation as the timed steps (first-use costs of the counting ops land there)
        warm_every = max(1, min(every, args.warmup // 2))  # at least two instrumented warm-up steps
        hook.profile_hop, hook.profile_every, hook.profile_log = last_hop, warm_every, []
        hook.profile_pool = [KernelTimer() for _ in range(args.warmup // warm_every + 1)]
        run(args.warmup)
        hook.check()
        hook.profile_every, hook.profile_log, hook._calls = every, [], 0
        # at most 48 timed launches: ~100 HIP events awaiting their timestamps is where the runtime starts to stall
        hook.profile_pool = [KernelTimer() for _ in range(min(48, steps // every + 1))]
        if real_world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        wake_host()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        if real_world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        hook.check()

    # ---- the same steps through the library's DEFAULT arguments (what an unmodified TGM script gets): DGDataLoader(dg, bs,
    # hook_manager=hm) + RecencyNeighborHook(...) with validate='sync'; the consumer holds batch i while batch i + 1 is produced,
    # like `for batch in loader` does ----
    default_elapsed = None
    if not args.no_default_path and args.mode == 'ring':
        dg2, hm2, hook2, loader2 = build_pipeline(stream, rank, world, global_bs, num_nbrs, args.mode, device, pool=None, validate=None)
        with hm2.activate('bench'):
            it, held = 0, None
            for _ in range(first_timed):
                held = loader2(starts[it])
                it += 1
            if real_world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()
            wake_host()
            t0 = time.perf_counter()
            for _ in range(steps):
                if it == n_batches:
                    hm2.reset_state()
                    it = 0
                held = loader2(starts[it])
                it += 1
            torch.cuda.synchronize()
            if real_world > 1:
                torch.distributed.barrier()
            default_elapsed = time.perf_counter() - t0
            # The same loop over a longer stretch, without a synchronize in between (what a pass over a data set sees; LATER batches, whose rows
            # hold a few per cent more neighbours).  Before wake_host() existed the K steps above measured the host's wake-up from the blocking
            # synchronize: 42-57 us in the first 20-step window, 36-41 in every later one (profiles/r06_default_path_windows.jsonl).
            steady_steps = max(100, steps)
            for _ in range(20):
                if it == n_batches:
                    hm2.reset_state()
                    it = 0
                held = loader2(starts[it])
                it += 1
            t0 = time.perf_counter()
            for _ in range(steady_steps):
                if it == n_batches:
                    hm2.reset_state()
                    it = 0
                held = loader2(starts[it])
                it += 1
            torch.cuda.synchronize()
            steady_elapsed = time.perf_counter() - t0
            hook2.check()
            default_sets = len(loader2._compiled[1]._sets) if loader2._compiled and loader2._compiled[1] is not None else 0
            # the same default arguments with the consumer DROPPING batch i before it asks for batch i + 1 (`del batch` at the end of the
            # loop body): the fresh-tensor bookkeeping then finds the one output set free again and stays on it -- the headline's cache
            # residency with the default arguments; the difference to the loop above is the second 177 MB set, not host time
            held = None
            torch.cuda.synchronize()
            wake_host()
            t0 = time.perf_counter()
            for _ in range(steps):
                if it == n_batches:
                    hm2.reset_state()
                    it = 0
                held = loader2(starts[it])
                it += 1
                held = None
            torch.cuda.synchronize()
            released_elapsed = time.perf_counter() - t0
            # host side alone: the same calls without waiting for the device in between are paced by the host if it is the bottleneck;
            # time.process_time() (CPU seconds of this process) over the region / steps = host busy time per step
            c0 = time.process_time()
            for _ in range(steps):
                if it == n_batches:
                    hm2.reset_state()
                    it = 0
                held = loader2(starts[it])
                it += 1
            host_busy = (time.process_time() - c0) / steps
            torch.cuda.synchronize()
            hook2.check()
            del held

    # ---- N > 1: proof that N ranks met over RCCL (the data path itself has no collective): ranks counted by an all-reduce,
    # and what a small all-reduce costs on this node's xGMI links ----
    rccl = None
    if real_world > 1:
        import torch.distributed as dist

        ones = torch.ones(1, dtype=torch.float32, device=device)
        dist.all_reduce(ones)
        buf = torch.zeros(1024, dtype=torch.float32, device=device)
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        rccl = {'backend': dist.get_backend(), 'ranks_seen': int(ones.item()), 'allreduce_us': 1e6 * (time.perf_counter() - t0) / 20,
                'what': '4 KiB float32 all-reduce, 20 back to back (latency of one collective on this node); not on the sampler\'s data path'}

    if real_world > 1:
        t = torch.tensor([elapsed, default_elapsed or 0.0], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0].item())
        default_elapsed = float(t[1].item()) if default_elapsed is not None else None

    # units actually processed by the timed steps (the ragged last batch of an epoch is smaller), all ranks together
    total_units, total_events, it = 0, 0, first_timed
    for _ in range(steps):
        if it == n_batches:
            it = 0
        n_e = min(global_bs, stream.num_edges - starts[it])
        total_events += n_e * (world if (by_batch and not args.emulate_world) else 1)
        for r in (range(world) if not args.emulate_world else [rank]):
            # by batch: every rank's step is a full batch of the schedule (the ragged last one is never reached by the default step count)
            total_units += slots_of(n_e if by_batch else (n_e * (r + 1)) // world - (n_e * r) // world)
        it += 1

    # ---- roofline of the dominant kernel (last hop's lookup + gather launch) -----------
    log = hook.profile_log
    hook.profile_hop = None
    if not log:
        raise SystemExit('bench.py: no launch of the dominant kernel was timed (steps too small for --profile-every?)')
    ls = launch_stats(log, D)
    avg_ms, ker_ms, seeds_l, total_slots, valid, delta, feat_slots, algo_bytes = (ls['avg_ms'], ls['ker_ms'], ls['seeds'], ls['slots'], ls['valid'],
                                                                                   ls['delta'], ls['feat_slots'], ls['algo_bytes'])
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
    shape = ls['shape']
    fused = len(shape) > 1
    # (which kernel serves the last hop alone: the narrow-row tile kernel from 2 tiles per CU on, the packed kernel below that, a wave per seed for wide rows)
    solo = ('recency_lookup_kernel' if D * num_nbrs[-1] > 1024 or max(num_nbrs) > 32 else
            ('lookup_tile_coop_kernel' if shape[-1][0] >= 2 * 64 * 256 and max(num_nbrs) <= 20 else ('lookup_tile_kernel' if shape[-1][0] >= 2 * 64 * 256 else 'lookup_packed_kernel')))
    kernel_name = ('recency_lookup_fused01_kernel (hop 0 + hop 1 in one launch: ' if fused else f'{solo} (hop {last_hop}: ') + \
        ' + '.join(f'{seeds} seeds x k={k}' for seeds, k in shape) + ')'

    lowered = args.pool > 0
    pkey = profile_key(args, bs, num_nbrs, steps, world)
    out = {
        'metric': 'sampled-edges/sec (TGAT 2-hop k=20 recency sampler, tgbl-wiki synthetic)' if args.workload == 'wiki'
        else f'sampled-edges/sec (recency sampler, tgbl-{args.workload} synthetic)',
        'value': total_units / elapsed,
        'unit': 'sampled-edges/s',
        'n_gpus': real_world,
        'steps': steps,
        'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / steps,
        'higher_is_better': True,
        'scaling': 'weak' if by_batch else args.scaling,  # by batch: every rank's timed region is `steps` full batches (per-GPU work fixed)
        'vs_baseline': None,
        'dtype': 'int32/int64 indices + f32 feature rows (copied, no arithmetic)',
        'data': 'synthetic',
        'config': {
            'workload': f'tgbl-{args.workload}-shaped synthetic stream: N={stream.num_nodes}, E={stream.num_edges}, D={D}; '
            f'seeds = src|dst|neg, num_nbrs={num_nbrs}, global batch {global_bs} edges, {bs_rank} seed edges per rank, mode={args.mode}; '
            f'timed batches {first_timed}..{first_timed + steps - 1} of {n_batches} (batches before them replayed untimed: rings in steady state); '
            + (f'loader output_pool={args.pool}: one tgmx_pipeline_step per batch into a ring of preallocated, persistent outputs '
               + ('(every neighbor is looked up and copied every batch; a feature row is rewritten from its leftmost slot that changes on -- the '
                  'all-pad slots left of it hold zeros already --, ids and times in full; TGMX_DELTA_WRITES=0 rewrites every slot); '
                  if os.environ.get('TGMX_DELTA_WRITES', '1') != '0' else '(every slot rewritten: TGMX_DELTA_WRITES=0); ') if lowered
               else 'hook-by-hook path, fresh output tensors per batch; ')
            + f"seed validation on the device, validate='{args.validate}'"
            + (' (status word read back once after the timed steps)' if args.validate == 'deferred' else ''),
            'slots_per_step_per_rank': slots_of(bs_rank),
            'profile_key': pkey,
            'events_per_s': total_events / elapsed,
            'parallelism': (f'batch-level sharding x{world}: the single-GPU schedule of {global_bs}-edge batches dealt round-robin to the ranks '
                            '(rank r takes batches r, r + N, ...; same batches, same sampled neighbours as one GPU), static index, replicated stream, '
                            'no data-path collective' if by_batch else
                            f'edge-batch sharding x{world} ({args.scaling} scaling), replicated stream, no data-path collective')
            + (f' -- EMULATED: this is rank {rank} of {world} alone on one GPU (value = that rank\'s units only)' if args.emulate_world else ''),
        },
        'roofline': {
            'bound': 'hbm',
            'kernel': kernel_name,
            'achieved': achieved,
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            'traffic': None,
            'avg_kernel_ms': avg_ms,
            'rocprof_avg_kernel_us': rocprof_kernel_us(pkey) if world == 1 else None,
            'launches_timed': len(ker_ms),
            'algorithmic_bytes_per_launch': algo_bytes,
            'valid_slot_fraction': valid / max(total_slots, 1),
            'feature_rows_written_fraction': feat_slots / max(total_slots, 1),
            'bytes_model': ('valid-aware, delta feature writes into the persistent output set: slots x 12 (ids, times) + rewritten slots x 4D '
                            '(a row is written from its first changing slot on: max(valid before, valid now)) + valid slots x (16 + 4D) read + 76 B per seed')
            if delta else 'valid-aware: slots x (12 + 4D) written + valid slots x (16 + 4D) read + 68 B per seed',
        },
    }
    # ---- what the headline fraction does NOT say, measured in this process on the same timed batches (N = 1, wiki) --------------
    extras = args.extras == 'on' or (args.extras == 'auto' and real_world == 1 and not args.emulate_world and args.workload == 'wiki' and args.mode == 'ring'
                                     and args.pool == 1)
    rl = out['roofline']
    rl['traffic'] = pmc_traffic(pkey, rl)
    if args.mode == 'ring' and 256 < 2 * bs_rank <= 1024 and len(num_nbrs) >= 2 and os.environ.get('TGMX_MERGE_LATE', '1') != '0' and 'packed' in kernel_name:
        # (two packed lookup launches, 512 < m <= 1024: the review-shaped step since round 6)
        rl['launch_carries'] = ("the ring update's merge + placement riders (a workgroup per 256-entry chunk, the last one out places): they, not the "
                                'lookups, set this launch\'s duration -- the lookups alone need ~8 us (TGMX_MERGE_LATE=0 puts sort + merge back into the hop-0 '
                                'launch: that one then lasts ~13 us and the step ~2 us longer)')
    rl['literal_8d_bytes'] = ls['literal_8d_bytes']
    rl['literal_8d_frac'] = ls['literal_8d_bytes'] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    rl['literal_8d_note'] = ('SURVEY 8(d) read literally: (28 + 8D) B per output slot + 68 B per seed.  It charges every PAD slot a 16-byte record read, a '
                             '4D-byte feature-row read and a feature-row write; pad slots (1 - valid_slot_fraction of them) read nothing, and with delta '
                             'writes most of them are not rewritten either, so this figure exceeds what the launch moves (traffic) and can exceed the HBM peak')
    rl['cache_note'] = ('wiki-shaped working set: ring_x 127 MB read + ONE 177 MB output set rewritten every batch (pool of one) -- mostly resident in the 256 MiB '
                        'Infinity Cache, and FETCH_SIZE / WRITE_SIZE count L2<->fabric bytes, not DRAM bytes: the fraction above is MALL-assisted.  '
                        'roofline_hbm_bound (comment-shaped, working set >> MALL) is the HBM-honest figure; variants.pool2 / full_write show this launch '
                        'without the cache residency / without delta writes')
    if extras:
        # (each block is best effort: a failure is recorded in the line, the contract's fields above are already final)
        def guarded(name, fn):
            try:
                fn()
            except Exception as exc:  # noqa: BLE001
                out.setdefault('extras_errors', {})[name] = f'{type(exc).__name__}: {exc}'[:300]
                torch.cuda.empty_cache()

        def _variants():
            n_probe = 64
            fw = probe_variant(stream, bs, num_nbrs, args.mode, device, first_timed, n_probe, pool=1, env={'TGMX_DELTA_WRITES': '0'})
            p2 = probe_variant(stream, bs, num_nbrs, args.mode, device, first_timed, n_probe, pool=2)
            rl['variants'] = {
                'full_write': {'avg_kernel_ms': fw['avg_ms'], 'algorithmic_bytes_per_launch': fw['algo_bytes'], 'frac': fw['algo_bytes'] / (fw['avg_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               'us_per_step': fw['us_per_step'], 'launches_timed': len(fw['ker_ms']),
                               'what': 'TGMX_DELTA_WRITES=0, pool of one: every slot of every feature row rewritten (slots x (12 + 4D) + valid x (16 + 4D) + 68 B per seed)'},
                'pool2': {'avg_kernel_ms': p2['avg_ms'], 'algorithmic_bytes_per_launch': p2['algo_bytes'], 'frac': p2['algo_bytes'] / (p2['avg_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          'us_per_step': p2['us_per_step'], 'launches_timed': len(p2['ker_ms']),
                          'what': 'output_pool=2, delta writes: two 177 MB output sets alternate, so a set has left the Infinity Cache when it is written again'},
            }

        def _hbm_bound():
            # the HBM-bound shape: comment-shaped stream (N = 1 M, E = 44 M, D = 16), bs 4096, k = [20, 20] -- working set >> 256 MiB MALL
            t_c = time.perf_counter()
            cs = make_stream('comment', seed=args.seed, device=device)
            cbs, cnb = DEFAULTS['comment']
            n_cb = (cs.num_edges + cbs - 1) // cbs
            hb = {}
            for cmode in ('ring', 'csr'):
                st_c = probe_variant(cs, cbs, cnb, cmode, device, n_cb // 2, 48, pool=1)
                hb[cmode] = {'kernel': 'lookup_tile_coop_kernel (hop 1: %d seeds x k=%d)' % (st_c['shape'][-1][0], st_c['shape'][-1][1]),
                             'achieved': st_c['algo_bytes'] / (st_c['avg_ms'] * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': st_c['algo_bytes'] / (st_c['avg_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 'avg_kernel_ms': st_c['avg_ms'],
                             'min_max_kernel_ms': [min(st_c['ker_ms']), max(st_c['ker_ms'])], 'launches_timed': len(st_c['ker_ms']),
                             'algorithmic_bytes_per_launch': st_c['algo_bytes'], 'valid_slot_fraction': st_c['valid'] / max(st_c['slots'], 1),
                             'us_per_step': st_c['us_per_step'], 'sampled_edges_per_s': slots_of_shape(cbs, cnb) / (st_c['us_per_step'] * 1e-6),
                             'timed_batches': f"{st_c['first_timed']}..{st_c['first_timed'] + 47} of {n_cb}"}
            del cs
            torch.cuda.empty_cache()
            out['roofline_hbm_bound'] = {'workload': 'tgbl-comment-shaped synthetic stream: N=1000000, E=44000000, D=16; bs=4096, k=[20, 20], pool of one, delta writes; '
                                                     'rings (mode=ring) and the static index (mode=csr); the narrow-row lookup of hop 1, same byte model as roofline',
                                         'bound': 'hbm', **{('static_index' if m == 'csr' else 'rings'): v for m, v in hb.items()},
                                         'seconds_spent': time.perf_counter() - t_c,
                                         'traffic': 'profiles/r0N_comment_{ring,csr}_pmc.json, newest round (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of `bench.py --workload comment`, separate runs)'}

        def _aggregation():
            out['aggregation'] = aggregation_block(stream, bs, num_nbrs, device, 100, first_timed)

        def _cfg3():
            # BASELINE cfg 3 (review-shaped stream, TGN memory + TransformerConv embedding, k = [10, 10], bs = 512): the whole per-batch pipeline
            # -- sampler -> dedup -> edge list -> memory -> embedding -> update_state -- as tools/bench_tgn.py measures it (its own process:
            # the script owns its graph and modules), with the loader's chain on its own stream beside the model's (DGDataLoader(side_stream=True),
            # DESIGN.md 3.3c) and on one stream
            import subprocess

            t_c = time.perf_counter()
            res = {}
            for name, env in (('two_streams', {}), ('one_stream', {'TGMX_BENCH_TGN_STREAMS': '0'})):
                r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_tgn.py'), '300'], env=dict(os.environ, TGMX_BENCH_TGN_NO_LOADER_PASS='1', TGMX_BENCH_TGN_REPEATS='3', **env),
                                   capture_output=True, text=True, timeout=240)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
                if r.returncode or not line:
                    raise RuntimeError(f'tools/bench_tgn.py ({name}) failed: {r.stderr[-200:]}')
                d = json.loads(line[-1])
                res[name] = {k: d[k] for k in ('pipeline_us_per_batch', 'host_busy_us_per_batch', 'host_waiting_for_the_device_us_per_batch', 'events_per_s',
                                               'sampled_edges_per_s', 'windows_us_per_batch')}
            out['pipeline_cfg3'] = {'what': 'BASELINE cfg 3: review-shaped synthetic (N = 350 k, E = 4.8 M, D = 16), TGN memory (Last, GRU, 100) + TransformerConv embedding, '
                                            'k = [10, 10], bs = 512; the same 300 batches after 100 of warm-up three times (state reset in between), the MEDIAN repeat reported (windows_us_per_batch: all three); '
                                            'tools/bench_tgn.py in its own process on this GPU', **res,
                                    'seconds_spent': time.perf_counter() - t_c}

        guarded('variants', _variants)
        guarded('roofline_hbm_bound', _hbm_bound)
        guarded('aggregation', _aggregation)
        guarded('pipeline_cfg3', _cfg3)
    if rccl is not None:
        out['rccl'] = rccl
    if real_world > 1:
        out['multi_gpu_note'] = (
            "the headline fields of this line are the wiki-shaped stream with STREAMING RINGS (the single-GPU default, so that the 1 -> N curve starts "
            "at BENCH's N = 1 figure): under N > 1 rings are REPLICAS -- every rank replays the whole global batch's ring update (N x bs edges; a chain "
            "that does not shrink with N), only the lookups are sharded.  The multi-GPU mode of this library is the static index "
            "(RecencyNeighborHook(mode='csr'): no per-batch state, no replicated update, no data-path collective): scale_comment below measures it on "
            "north_star's scaling shape in this same job, batch-sharded (DGDataLoader(batch_shard=)) and weak (EdgeShardHook).")
    if real_world > 1 and args.scale_comment != 'off':
        try:
            out['scale_comment'] = scale_comment_block(args, rank, world, device)
            out['scale_comment']['rccl_ranks_seen'] = rccl['ranks_seen']
        except Exception as exc:  # noqa: BLE001 -- every rank takes the same path (the block's collectives are symmetric)
            out['scale_comment'] = {'error': f'{type(exc).__name__}: {exc}'[:300]}
    if real_world > 1 and args.tgn_allgather != 'off':
        try:
            out['tgn_memory_allgather'] = tgn_memory_allgather_block(args, rank, world, device)
        except Exception as exc:  # noqa: BLE001 -- symmetric across ranks, like scale_comment
            out['tgn_memory_allgather'] = {'error': f'{type(exc).__name__}: {exc}'[:300]}
    out['valid_edges_per_s'] = out['value'] * out['roofline']['valid_slot_fraction']  # sampled slots that hold a neighbor (pads excluded)
    if default_elapsed is not None:
        out['default_path'] = {
            'ms_per_step': 1e3 * default_elapsed / steps,
            'value': total_units / default_elapsed,
            'what': "the same timed steps through DGDataLoader(dg, batch_size, hook_manager=hm) and RecencyNeighborHook(...) with their DEFAULT "
            "arguments (validate='sync': raise-per-call; fresh-tensor semantics: an output set is reused only once nothing can reach its "
            f'tensors), the consumer holding batch i while batch i + 1 is produced like `for batch in loader`; {default_sets} output sets in use',
            'steady_ms_per_step': 1e3 * steady_elapsed / steady_steps,
            'steady_what': f'the same held-batch loop over the next {steady_steps} steps, started 20 steps after the timed region without a synchronize in between '
                           '(LATER batches of the stream, whose rows hold more neighbours: a few per cent more bytes per step)',
            'released_ms_per_step': 1e3 * released_elapsed / steps,
            'released_what': 'the same, the consumer dropping batch i before asking for batch i + 1 (one output set, which stays in the Infinity Cache like '
                             "the headline's pool of one): the gap between the two figures is the second 177 MB output set leaving the cache, not host time",
            'host_busy_us_per_step': 1e6 * host_busy,
        }
    if rank == 0 or args.emulate_world:
        if world == 1 and real_world == 1 and args.cpu_batches > 0:
            out['cpu_baseline'] = cpu_baseline(stream, bs, num_nbrs, args.cpu_batches, args.seed, first_timed)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if real_world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

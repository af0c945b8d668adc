#!/usr/bin/env python
"""Rank 0's sampler step of a W-rank job, timed on ONE GPU for W in {1, 2, 4, 8} (no process group: the data path has
no collective, so rank 0's work -- its 200-edge shard of the lookups + the whole W x 200-edge global batch's ring
update -- is exactly what it would do next to W - 1 peers).  Prints wall time per step (host and GPU overlapped) and
the GPU-only time per step (events around the same loop)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import DEFAULTS, build_pipeline  # noqa: E402
from tgm_amd.synth import make_stream  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'wiki'
mode = sys.argv[2] if len(sys.argv) > 2 else 'ring'
worlds = tuple(int(w) for w in sys.argv[3].split(',')) if len(sys.argv) > 3 else (1, 2, 4, 8)
steps = int(os.environ.get('TGMX_STEPS', '600'))
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
bs_rank, num_nbrs = DEFAULTS[workload]
stream = make_stream(workload, seed=1337, device='cpu' if workload == 'wiki' else dev)
out = {}
for world in worlds * int(os.environ.get('TGMX_ROUNDS', '2')):
    dg, hm, hook, loader = build_pipeline(stream, 0, world, bs_rank, num_nbrs, mode, dev)
    starts = loader._starts
    nb = len(loader)
    with hm.activate('bench'):
        it = 0
        best = None
        for rep in range(4):  # first repetition = warm-up
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            for _ in range(steps):
                if it == nb:
                    hm.reset_state()
                    it = 0
                loader(starts[it])
                it += 1
            e1.record()
            host = (time.perf_counter() - t0) / steps * 1e6  # enqueue only: below the wall time when the GPU is the bound
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / steps * 1e6
            gpu = e0.elapsed_time(e1) / steps * 1e3
            if rep and (best is None or wall < best[0]):
                best = (wall, gpu, host)
        hook.check()
    prev = out.get(world)
    if prev is None or best[0] < prev['wall_us_per_step']:
        out[world] = {'wall_us_per_step': round(best[0], 1), 'stream_us_per_step': round(best[1], 1), 'host_enqueue_us_per_step': round(best[2], 1)}
    print(world, out[world], flush=True)
base = out[worlds[0]]['wall_us_per_step']
print(json.dumps({'workload': workload, 'mode': mode, 'rank0_step_by_world': out,
                  'modelled_weak_scaling_efficiency': {w: round(base / out[w]['wall_us_per_step'], 3) for w in out}}))

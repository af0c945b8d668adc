#!/usr/bin/env python
"""Where the host's microseconds go in the DEFAULT-argument path (DGDataLoader(dg, bs, hook_manager=hm), RecencyNeighborHook(...)
with validate='sync', fresh-tensor semantics), the consumer holding batch i while batch i + 1 is produced.  Run on the GPU box.
  python tools/host_profile_default.py [held|released] [pool]     (pool: an explicit output_pool instead of the default None)"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tgm_amd.synth import make_stream  # noqa: E402

held_mode = (sys.argv[1] if len(sys.argv) > 1 else 'held') == 'held'
pool = int(sys.argv[2]) if len(sys.argv) > 2 else None
stream = make_stream('wiki', seed=1337)
dev = torch.device('cuda', 0)
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, 200, [20, 20], 'ring', dev, pool=pool, validate=None if pool is None else 'deferred')
starts = loader._starts
with hm.activate('bench'):
    held = None
    for i in range(400):
        held = loader(starts[i])
        if not held_mode:
            held = None
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        c0 = time.process_time()
        for i in range(400, 700):
            held = loader(starts[i])
            if not held_mode:
                held = None
        c1 = time.process_time()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'{"held" if held_mode else "released"} pool={pool}: host {1e6 * (t1 - t0) / 300:.1f} us/step (cpu {1e6 * (c1 - c0) / 300:.1f}), with drain {1e6 * (t2 - t0) / 300:.1f} us/step', flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(100, 600):
        held = loader(starts[i])
        if not held_mode:
            held = None
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)

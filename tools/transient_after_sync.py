#!/usr/bin/env python
"""Host microseconds of every loader call in the first steps after a torch.cuda.synchronize(), next to the same calls once the host runs ahead
of the device.   python tools/transient_after_sync.py [headline|default] [spin_ms]   (headline: pool of one, deferred validation; default: the loader's and the
hook's default arguments, the consumer holding batch i while batch i + 1 is produced)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tgm_amd.synth import make_stream  # noqa: E402

stream = make_stream('wiki', seed=1337)
dev = torch.device('cuda', 0)
default = len(sys.argv) > 1 and sys.argv[1] == 'default'
spin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, 200, [20, 20], 'ring', dev, pool=None if default else 1, validate=None if default else 'deferred')
starts = loader._starts
res = []
with hm.activate('bench'):
    it = 0
    for _ in range(399):
        held = loader(starts[it]); it += 1
    for rep in range(4):
        torch.cuda.synchronize()
        if spin_ms:  # keep the core awake: a blocking synchronize parks the thread, and the core comes back slow for ~1 ms
            t_s = time.perf_counter()
            while time.perf_counter() - t_s < spin_ms * 1e-3:
                pass
        per = []
        t0 = time.perf_counter()
        for _ in range(60):
            a = time.perf_counter()
            held = loader(starts[it]); it += 1
            per.append(round(1e6 * (time.perf_counter() - a), 1))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append({'host_us_per_call': per, 'wall_us_per_step': round(1e6 * (t2 - t0) / 60, 1), 'host_total_us_per_step': round(1e6 * (t1 - t0) / 60, 1)})
    hook.check()
print(json.dumps(res))

#!/usr/bin/env python
"""Host microseconds of every loader call in the first steps after a torch.cuda.synchronize() (headline pipeline: pool of one, deferred
validation), next to the same calls once the host runs ahead of the device.   python tools/transient_after_sync.py"""
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
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, 200, [20, 20], 'ring', dev, pool=1)
starts = loader._starts
res = []
with hm.activate('bench'):
    it = 0
    for _ in range(399):
        loader(starts[it]); it += 1
    for rep in range(4):
        torch.cuda.synchronize()
        per = []
        t0 = time.perf_counter()
        for _ in range(40):
            a = time.perf_counter()
            loader(starts[it]); it += 1
            per.append(round(1e6 * (time.perf_counter() - a), 1))
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append({'host_us_per_call': per, 'wall_us_per_step_40': round(1e6 * (t2 - t0) / 40, 1), 'host_total_us_per_step': round(1e6 * (t1 - t0) / 40, 1)})
    hook.check()
print(json.dumps(res))

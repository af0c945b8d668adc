#!/usr/bin/env python
"""Why does `default_path` (20 timed steps) read 37 us/step in some bench runs and 43-47 in others?  Times consecutive 20-step windows of the
held-batch loop through the DEFAULT arguments, in a fresh process, (a) alone, (b) with a second pipeline's state (rings + one output set,
what bench.py's headline region leaves alive) resident beside it.   python tools/default_path_windows.py [alone|beside] [windows]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tgm_amd.synth import make_stream  # noqa: E402

beside = (sys.argv[1] if len(sys.argv) > 1 else 'alone') == 'beside'
n_win = int(sys.argv[2]) if len(sys.argv) > 2 else 12
stream = make_stream('wiki', seed=1337)
dev = torch.device('cuda', 0)
keep = None
if beside:
    dg0, hm0, hook0, loader0 = bench.build_pipeline(stream, 0, 1, 200, [20, 20], 'ring', dev, pool=1)
    with hm0.activate('bench'):
        for i in range(420):
            loader0(loader0._starts[i])
    torch.cuda.synchronize()
    keep = (dg0, hm0, hook0, loader0)
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, 200, [20, 20], 'ring', dev, pool=None, validate=None)
starts = loader._starts
out = []
with hm.activate('bench'):
    held, it = None, 0
    for _ in range(399):
        held = loader(starts[it]); it += 1
    torch.cuda.synchronize()
    for w in range(n_win):
        t0 = time.perf_counter()
        for _ in range(20):
            held = loader(starts[it]); it += 1
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append((round(1e6 * (t2 - t0) / 20, 1), round(1e6 * (t1 - t0) / 20, 1)))
    hook.check()
print(json.dumps({'mode': 'beside' if beside else 'alone', 'us_per_step_wall_and_host_per_20_step_window': out,
                  'sets': len(loader._compiled[1]._sets)}))

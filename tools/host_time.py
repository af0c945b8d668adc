#!/usr/bin/env python
"""Host vs device side of one sampler step: time to ENQUEUE n steps (no sync inside) and time until the device has
finished them, plus a cProfile of the enqueue loop.   python tools/host_time.py [--workload review] [--pool 1] [--steps 400]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='wiki')
ap.add_argument('--pool', type=int, default=1)
ap.add_argument('--steps', type=int, default=400)
ap.add_argument('--mode', default='ring')
ap.add_argument('--profile', action='store_true')
a = ap.parse_args()
from tgm_amd.synth import make_stream  # noqa: E402

dev = torch.device('cuda', 0)
bs, ks = bench.DEFAULTS[a.workload]
st = make_stream(a.workload, seed=1337, device='cpu' if a.workload == 'wiki' else dev)
dg, hm, hook, loader = bench.build_pipeline(st, 0, 1, bs, ks, a.mode, dev, pool=a.pool)
starts = loader._starts
with hm.activate('bench'):
    for i in range(len(starts) // 2):
        loader(starts[i])
    torch.cuda.synchronize()
    i0 = len(starts) // 2
    a.steps = min(a.steps, (len(starts) - i0) // 4)  # three timed repetitions (+ the profiled one) walk on through the stream
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(i0, i0 + a.steps):
            loader(starts[i])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'{a.workload} pool={a.pool} mode={a.mode}: enqueue {1e6 * (t1 - t0) / a.steps:.1f} us/step, until done {1e6 * (t2 - t0) / a.steps:.1f} us/step', flush=True)
        i0 += a.steps
    if a.profile:
        pr = cProfile.Profile()
        pr.enable()
        for i in range(i0, i0 + a.steps):
            loader(starts[i])
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats('tottime').print_stats(14)

#!/usr/bin/env python
"""cProfile of the per-batch host path (run on the GPU box): where do the microseconds go?"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tgm_amd.synth import make_stream  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'ring'
stream = make_stream('wiki', seed=1337)
dev = torch.device('cuda', 0)
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, 200, [20, 20], mode, dev)
starts = loader._starts
with hm.activate('bench'):
    for i in range(30):
        loader(starts[i])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(30, 530):
        loader(starts[i])
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)

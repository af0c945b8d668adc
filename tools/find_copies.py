#!/usr/bin/env python
"""Where do the device copies of a cfg-3 batch come from?  torch.profiler with stacks over a few batches of tools/bench_tgn.py's loop."""
import os, sys, runpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['TGMX_BENCH_TGN_NO_LOADER_PASS'] = '1'
from torch.profiler import profile, ProfilerActivity
sys.argv = ['bench_tgn.py', '60']
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_tgn.py'), run_name='__main__')
ev = [e for e in prof.events() if e.name in ('aten::copy_', 'aten::_to_copy', 'aten::contiguous', 'aten::clone', 'aten::cat', 'aten::index_select', 'aten::zeros', 'aten::zero_', 'aten::fill_')]
from collections import Counter
c = Counter()
for e in ev:
    st = [s for s in (e.stack or []) if 'tgm_amd' in s or 'bench_tgn' in s]
    c[(e.name, str(e.input_shapes)[:60], st[0] if st else '?')] += 1
for k, v in c.most_common(40):
    print(v, k)

#!/usr/bin/env python
"""TGAT training step (forward(save) + backward + fused Adam on one fixed headline batch) A/B INSIDE one process: per-call knobs of the library are
switched between segments of 30 steps (HIP events around each segment).  usage: train_inprocess_ab.py ROUNDS "A=1" "B=0" ...  ("-" = defaults)"""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tgm_amd.nn import TGAT  # noqa: E402
from tgm_amd.synth import make_stream  # noqa: E402

rounds = int(sys.argv[1])
cfgs = sys.argv[2:]
keys = sorted({kv.split('=')[0] for c in cfgs for kv in c.split() if '=' in kv})
stream = make_stream('wiki', seed=1337)
dev = torch.device('cuda', 0)
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, 200, [20, 20], 'ring', dev, pool=None, validate=None, edge_features='by_id')
starts = loader._starts
node_x = dg.static_node_x
torch.manual_seed(0)
enc = TGAT(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2).to(dev).train()
opt = torch.optim.Adam(enc.parameters(), lr=1e-4, fused=True)


def step(b):
    opt.zero_grad(set_to_none=True)
    z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
    pos = (z[:200] * z[200:400]).sum(-1)
    neg = (z[:200] * z[400:]).sum(-1)
    loss = torch.nn.functional.softplus(-pos).mean() + torch.nn.functional.softplus(neg).mean()
    loss.backward()
    opt.step()
    return loss


res = {c: [] for c in cfgs}
with hm.activate('bench'):
    for i in range(301):
        b = loader(starts[i])
    for _ in range(30):
        step(b)
    torch.cuda.synchronize()
    for r in range(rounds):
        for c in (cfgs if r % 2 == 0 else cfgs[::-1]):
            for k in keys:
                os.environ.pop(k, None)
            for kv in c.split():
                if '=' in kv:
                    k, v = kv.split('=')
                    os.environ[k] = v
            for _ in range(3):
                step(b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                step(b)
            e1.record()
            torch.cuda.synchronize()
            res[c].append(e0.elapsed_time(e1) / 30 * 1000)
for c in cfgs:
    v = res[c]
    print(json.dumps({'config': c, 'median_us_per_step': round(statistics.median(v), 1), 'min': round(min(v), 1), 'all': [round(x, 1) for x in v]}))

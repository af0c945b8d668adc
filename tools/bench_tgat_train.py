#!/usr/bin/env python
"""Time one TGAT training step (forward with saved intermediates + hand-written backward + Adam) at the headline batch
shape on the GPU box; prints one JSON line.  `python tools/bench_tgat_train.py [n_steps]`"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tgm_amd.nn import TGAT  # noqa: E402
from tgm_amd.synth import make_stream  # noqa: E402

stream = make_stream('wiki', seed=1337)
dev = torch.device('cuda', 0)
features = sys.argv[2] if len(sys.argv) > 2 else 'dense'  # 'by_id': the sampler publishes edge ids, the attention kernels read the resident store
dg, hm, hook, loader = bench.build_pipeline(stream, 0, 1, 200, [20, 20], 'ring', dev, pool=None, validate=None, edge_features=features)  # the library's default loader / hook arguments
starts = loader._starts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
node_x = dg.static_node_x


def run(fused):
    """one optimizer variant: torch.optim.Adam as the reference example constructs it (foreach kernels), or fused=True (one kernel;
    its updates do not bump Tensor._version: tgm_amd.nn._paramver)"""
    torch.manual_seed(0)
    enc = TGAT(node_dim=1, edge_dim=172, time_dim=100, embed_dim=172, num_layers=2).to(dev).train()  # the reference default dropout 0.1
    opt = torch.optim.Adam(enc.parameters(), lr=1e-4, fused=fused)

    def step(b):
        opt.zero_grad(set_to_none=True)
        z = enc(node_x, b.seed_nids, b.seed_times, b.nbr_nids, b.nbr_edge_x, b.nbr_edge_time)
        pos = (z[:200] * z[200:400]).sum(-1)
        neg = (z[:200] * z[400:]).sum(-1)
        loss = torch.nn.functional.softplus(-pos).mean() + torch.nn.functional.softplus(neg).mean()
        loss.backward()
        opt.step()
        return loss

    b = loader(starts[run.at])
    for _ in range(20):  # (allocator and weight-layout warm-up)
        step(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        step(b)
    e1.record()
    torch.cuda.synchronize()
    fixed_us = e0.elapsed_time(e1) / 30 * 1000
    t0 = time.perf_counter()
    for i in range(run.at + 1, run.at + 1 + n):
        b = loader(starts[i])
        loss = step(b)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run.at += n + 1
    return {'train_step_us_fixed_batch': fixed_us, 'sampler_plus_train_step_us_per_batch': 1e6 * (t1 - t0) / n, 'edges_per_s': 200 * n / (t1 - t0),
            'final_loss': float(loss.detach())}  # fmt: skip


with hm.activate('bench'):
    for i in range(300):
        loader(starts[i])
    run.at = 300
    plain = run(None)
    fused = run(True)
print(json.dumps({
    'what': 'TGAT training step (sampler + forward(save) + backward + Adam), example dims, 600 seeds, k=[20,20], dot-product link loss',
    'backward': os.environ.get('TGMX_TGAT_BWD', 'native (tgmx_tgat_backward)'), 'edge_features': features,
    **plain, 'adam_fused': fused,
}))

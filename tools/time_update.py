#!/usr/bin/env python
"""Microbenchmark of tgmx_ring_update at several batch sizes (wiki-shaped ids, wrapping keys)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tgm_amd import _native

dev = torch.device('cuda', 0)
lib = _native.load()
N, B, D = 9227, 20, 172
g = torch.Generator().manual_seed(0)
out = {}
for n in [200, 400, 512, 800, 1600, 2048, 4096, 32768]:
    m = 2 * n
    src = torch.randint(0, 8227, (n,), generator=g).int().to(dev)
    dst = (8227 + (torch.rand(n, generator=g) ** 3 * 1000).long().clamp(max=999)).int().to(dev)
    ts = torch.sort(torch.randint(1_000_000, 1_003_000, (n,), generator=g)).values.to(dev)
    x = torch.rand(n, D, generator=g).to(dev)
    ring = torch.zeros(N * B, 2, dtype=torch.int64, device=dev)
    wpos = torch.zeros(N, dtype=torch.int32, device=dev)
    ring_x = torch.zeros(N * B, D, device=dev)
    scratch = torch.zeros(int(lib.tgmx_ring_update_scratch_bytes(n, 0)), dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    st = _native.stream_ptr()
    def call():
        rc = lib.tgmx_ring_update(ring.data_ptr(), wpos.data_ptr(), ring_x.data_ptr(), D, B, N, src.data_ptr(), dst.data_ptr(),
                                  ts.data_ptr(), x.data_ptr(), n, 0, 0, 1, scratch.data_ptr(), status.data_ptr(), st)
        assert rc == 0
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        call()
    e1.record()
    torch.cuda.synchronize()
    out[m] = round(e0.elapsed_time(e1) / 200 * 1e3, 2)
print(json.dumps({'ring_update_us_by_m': out}))

#!/usr/bin/env python
"""What this chip sustains for the traffic mixes of the gather kernels (torch ops on 1 GiB buffers, far beyond the 256 MiB
Infinity Cache): pure write (fill), copy (1 read : 1 write), and a 1 : 2.5 read : write mix like the sampler's output-heavy
launches (read 0.4 GiB, write 1 GiB).  Prints GB/s of total HBM traffic per case."""
import time

import torch

dev = torch.device('cuda', 0)
n = 1 << 28  # floats: 1 GiB
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
small = torch.empty(n * 2 // 5, dtype=torch.float32, device=dev)


def t(fn, bytes_, name, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f'{name}: {bytes_ / dt / 1e9:.0f} GB/s ({dt * 1e6:.0f} us)', flush=True)


t(lambda: a.zero_(), 4 * n, 'fill 1 GiB (write only)')
t(lambda: b.copy_(a), 8 * n, 'copy 1 GiB -> 1 GiB (1 read : 1 write)')
t(lambda: torch.sum(a), 4 * n, 'reduce 1 GiB (read only)')
idx = torch.arange(n, device=dev) % small.numel()
t(lambda: torch.index_select(small, 0, idx, out=b), 4 * n + 4 * small.numel() + 8 * n, 'gather 0.4 GiB -> 1 GiB (+ 2 GiB of int64 indices read)')
m = 200 * 1024 * 1024 // 4
t(lambda: a[:m].zero_(), 4 * m, 'fill 200 MiB (fits the Infinity Cache)')
t(lambda: b[:m].copy_(a[:m]), 8 * m, 'copy 200 MiB')

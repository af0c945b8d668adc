#!/usr/bin/env python
"""Host cost of cross-stream ordering on this box: event record + stream wait per iteration with small kernels in flight on both streams
(what a two-stream pipeline -- loader chain beside the model's chain -- would add per batch)."""
import time, torch
dev = torch.device('cuda', 0)
a = torch.zeros(1 << 16, device=dev); b = torch.zeros(1 << 16, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(64)]
def loop(n, sync):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(s1):
            for _ in range(4): a.add_(1.0)
            if sync:
                e = evs[(2 * i) % 64]; e.record(s1)
        with torch.cuda.stream(s2):
            if sync: s2.wait_event(e)
            for _ in range(4): b.add_(1.0)
            if sync:
                f = evs[(2 * i + 1) % 64]; f.record(s2)
        if sync: s1.wait_event(f)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n
for sync in (False, True, False, True):
    h, w = loop(2000, sync)
    print(f'sync={sync}: host {h:.1f} us / iteration, wall {w:.1f} us / iteration (8 tiny kernels per iteration' + (', 2 records + 2 waits)' if sync else ')'))

#!/usr/bin/env python
"""Kernel timeline of the last few steps of a rocprofv3 --kernel-trace run (start / end relative to the step's first launch)."""
import glob, os, sqlite3, sys

path = sys.argv[1]
anchor = sys.argv[2]  # substring of the kernel that opens a step
dbs = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))
db = sqlite3.connect(dbs[-1])
rows = db.execute('select name, start, end, grid_x, queue_id from kernels order by start').fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
if len(idx) < 4:  # no such kernel: show the last launches as they come
    lo, hi = max(0, len(rows) - int(sys.argv[3]) - 1 if len(sys.argv) > 3 else len(rows) - 25), len(rows) - 1
else:
    lo, hi = idx[-5], idx[-3]
t0 = rows[lo][1]
for name, s, e, g, q in rows[lo:hi]:
    short = name.split('(')[0].replace('void ', '').replace('tgmx::', '')[:60]
    print(f'{(s - t0) / 1e3:8.1f} {(e - t0) / 1e3:8.1f}  {(e - s) / 1e3:7.1f} us  q{q}  grid {g:>9}  {short}')
print(f'step wall: {(rows[hi][1] - t0) / 1e3:.1f} us')

#!/usr/bin/env python
"""Summarise rocprofv3 output (the rocpd SQLite database this image's rocprofv3 writes) as markdown / JSON.

    python tools/prof_summary.py trace <dir-or-db> [--title "..."]      per-kernel table of a --kernel-trace --stats run
    python tools/prof_summary.py byname <dir-or-db> [--steps N]         the same grouped by kernel name only (+ launches / us per step)
    python tools/prof_summary.py gaps  <dir-or-db> --kernel <substr>    launch-to-launch gaps on the stream around one kernel
    python tools/prof_summary.py pmc   <dir-or-db> [--kernel <substr>]  per-kernel average of every collected counter

Used by tools/gpu_profile.sh; the tables under profiles/ are its output, unedited.
"""
import argparse
import glob
import json
import os
import sqlite3
import sys


def open_db(path):
    if os.path.isdir(path):
        dbs = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))
        if not dbs:
            sys.exit(f'no .db under {path}')
        path = dbs[-1]
    return sqlite3.connect(path)


def short(name, n=70):
    return name if len(name) <= n else name[: n - 3] + '...'


def trace(db, title):
    rows = db.execute('select name, grid_x, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size) '
                      'from kernels group by name, grid_x order by sum(duration) desc').fetchall()
    total = sum(r[3] for r in rows) or 1
    print(f'# {title}')
    print('| kernel | grid | calls | total ms | avg us | min us | max us | VGPR | LDS B | % |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
    for name, grid, n, tot, avg, mn, mx, vg, lds in rows[:30]:
        print(f'| `{short(name)}` | {grid} | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {vg} | {lds} | {100 * tot / total:.1f} |')


def byname(db, title, steps):
    """per-kernel-NAME table (all grid sizes of one kernel together): launches and microseconds per step when --steps is given.  For pipelines
    whose launch shapes change from batch to batch (TGN: the number of unique nodes), where the per-(name, grid) table is mostly tail."""
    rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size) '
                      'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    n_all = sum(r[1] for r in rows)
    print(f'# {title}')
    if steps:
        print(f'# {n_all} launches, {total / 1e6:.3f} ms of kernels over {steps} steps: {n_all / steps:.1f} launches and {total / 1e3 / steps:.1f} us of kernel time per step')
    print('| kernel | calls | per step | total ms | us per step | avg us | min us | max us | VGPR | LDS B | % |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
    for name, n, tot, avg, mn, mx, vg, lds in rows[:40]:
        ps = f'{n / steps:.2f}' if steps else ''
        us = f'{tot / 1e3 / steps:.2f}' if steps else ''
        print(f'| `{short(name)}` | {n} | {ps} | {tot / 1e6:.3f} | {us} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {vg} | {lds} | {100 * tot / total:.1f} |')


def gaps(db, kernel):
    rows = db.execute('select name, start, end from kernels order by start').fetchall()
    idle, busy, n = 0, 0, 0
    first = last = None
    for i, (name, s, e) in enumerate(rows):
        if kernel in name:
            if first is None:
                first = i
            last = i
    if first is None:
        sys.exit(f'no kernel matching {kernel}')
    # the timed steps are the second half of the run (bench.py replays the first half untimed): launches 55 % .. 95 %
    idx = [i for i, r in enumerate(rows) if kernel in r[0]]
    lo, hi = idx[int(0.55 * len(idx))], idx[int(0.95 * len(idx))]
    for i in range(lo, hi):
        busy += rows[i][2] - rows[i][1]
        idle += max(0, rows[i + 1][1] - rows[i][2])
        n += 1
    steps = sum(1 for i in idx if lo <= i < hi)
    print(json.dumps({'kernel': kernel, 'launches_in_window': n, 'steps_in_window': steps, 'busy_us_per_step': busy / steps / 1e3,
                      'idle_us_per_step': idle / steps / 1e3, 'wall_us_per_step': (rows[hi][1] - rows[lo][1]) / steps / 1e3}))


def tail(db, kernel, last):
    """average duration of the LAST `last` launches of one kernel: bench.py's timed steps are the end of the run"""
    rows = [r for r in db.execute('select name, start, end, grid_x from kernels order by start').fetchall() if kernel in r[0]]
    main_grid = max(set(r[3] for r in rows), key=lambda g: (sum(1 for r in rows if r[3] == g), g))  # (ties: the larger grid = the last hop)
    rows = [r for r in rows if r[3] == main_grid][-last:]
    d = [(e - s_) / 1e3 for _, s_, e, _ in rows]
    print(f'| timed region: last {len(d)} launches of `{kernel}` | avg {sum(d) / len(d):.2f} us | min {min(d):.2f} | max {max(d):.2f} |')


def pmc(db, kernel):
    rows = db.execute('select k.name, k.grid_x, p.counter_name, count(*), avg(p.counter_value), min(p.counter_value), max(p.counter_value) '
                      'from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id group by k.name, k.grid_x, p.counter_name '
                      'order by avg(p.counter_value) desc').fetchall()
    print('| kernel | counter | grid | n | avg | min | max |')
    print('|---|---|---:|---:|---:|---:|---:|')
    for name, grid, cn, n, avg, mn, mx in rows:
        if kernel and kernel not in name:
            continue
        print(f'| `{short(name, 60)}` | {cn} | {grid} | {n} | {avg:.1f} | {mn:.1f} | {mx:.1f} |')


def union(db, last_frac):
    """sum of kernel durations vs the length of the UNION of their [start, end) intervals over the last `last_frac` of the run's launches
    (two-stream pipelines: how much of the kernel time actually overlaps on the device), and the wall span they cover"""
    rows = db.execute('select start, end from kernels order by start').fetchall()
    rows = rows[int(len(rows) * (1.0 - last_frac)):]
    total = sum(e - s for s, e in rows)
    cover, cur_s, cur_e = 0, None, None
    for s, e in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                cover += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    cover += cur_e - cur_s
    span = max(e for _, e in rows) - rows[0][0]
    print(json.dumps({'launches': len(rows), 'sum_of_kernel_ms': total / 1e6, 'union_ms': cover / 1e6, 'span_ms': span / 1e6,
                      'overlap_saved_frac': 1 - cover / total, 'device_busy_frac_of_span': cover / span}))


def pmcsum(db, kernel):
    """per kernel NAME and counter: the per-dispatch SUM over the counter's instances (XCDs / shader engines), averaged over the dispatches; JSON lines"""
    rows = db.execute('select k.name, k.dispatch_id, p.counter_name, sum(p.counter_value), count(*) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id '
                      'group by k.name, k.dispatch_id, p.counter_name').fetchall()
    acc = {}
    for name, _, c, v, n in rows:
        if kernel and kernel not in name:
            continue
        acc.setdefault((name, c), []).append((v, n))
    for (name, c), vs in sorted(acc.items()):
        print(json.dumps({'kernel': short(name, 80), 'counter': c, 'dispatches': len(vs), 'avg_sum_per_dispatch': sum(v for v, _ in vs) / len(vs),
                          'instances_per_dispatch': vs[0][1]}))


def pmctail(db, kernel, last):
    """per-counter average over the LAST `last` dispatches of one kernel (bench.py's timed steps are the end of the run: the
    whole-run average also covers the untimed ring fill, whose launches move far fewer bytes); one JSON object"""
    rows = db.execute('select k.dispatch_id, k.start, k.grid_x, p.counter_name, sum(p.counter_value) from pmc_events p join kernels k '
                      'on k.dispatch_id = p.dispatch_id where k.name like ? group by k.dispatch_id, p.counter_name order by k.dispatch_id', ('%' + kernel + '%',)).fetchall()
    if not rows:
        sys.exit(f'no dispatch of a kernel matching {kernel}')
    grids = {}
    for d, _, g, _, _ in rows:
        grids.setdefault(g, set()).add(d)
    main_grid = max(grids, key=lambda g: (len(grids[g]), g))  # (ties: the larger grid = the last hop)
    ids = sorted(grids[main_grid])[-last:]
    keep = set(ids)
    out = {}
    for d, _, g, c, v in rows:
        if d in keep:
            out.setdefault(c, []).append(v)
    print(json.dumps({'kernel': kernel, 'grid_x': main_grid, 'dispatches': len(ids),
                      'counters': {c: {'avg': sum(v) / len(v), 'min': min(v), 'max': max(v)} for c, v in out.items()}}))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('what', choices=['trace', 'byname', 'gaps', 'pmc', 'pmcsum', 'tail', 'pmctail', 'union'])
    ap.add_argument('--last', type=int, default=393)
    ap.add_argument('path')
    ap.add_argument('--title', default='rocprofv3 --kernel-trace --stats')
    ap.add_argument('--kernel', default='')
    ap.add_argument('--steps', type=int, default=0)
    a = ap.parse_args()
    db = open_db(a.path)
    {'trace': lambda: trace(db, a.title), 'byname': lambda: byname(db, a.title, a.steps), 'gaps': lambda: gaps(db, a.kernel), 'pmc': lambda: pmc(db, a.kernel), 'pmcsum': lambda: pmcsum(db, a.kernel), 'union': lambda: union(db, 0.5), 'tail': lambda: tail(db, a.kernel, a.last),
     'pmctail': lambda: pmctail(db, a.kernel, a.last)}[a.what]()

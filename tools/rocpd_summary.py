#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (SQLite) result: per-kernel call count / total / average
duration (kernel trace) and, when present, per-kernel PMC counter averages.

    python tools/rocpd_summary.py <results.db> [--md]
"""
import sqlite3
import sys


def short(name: str, n: int = 78) -> str:
    name = name.replace('void ', '')
    return name if len(name) <= n else name[: n - 3] + '...'


def main():
    path = sys.argv[1]
    md = '--md' in sys.argv
    db = sqlite3.connect(path)
    cur = db.cursor()
    if '--xyz' in sys.argv:  # one line per launch shape
        q = "select name || ' [grid ' || grid_x || 'x' || grid_y || 'x' || grid_z || ']', count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc"
    else:
        q = "select name || ' [grid ' || grid_x || ']', count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name, grid_x order by sum(duration) desc"
    rows = cur.execute(q).fetchall()
    total = sum(r[2] for r in rows) or 1
    if rows:
        print('| kernel | calls | total ms | avg us | min us | max us | % |' if md else f'{"kernel":80s} {"calls":>7s} {"total ms":>10s} {"avg us":>9s} {"min us":>9s} {"max us":>9s} {"%":>6s}')
        if md:
            print('|---|---:|---:|---:|---:|---:|---:|')
        for name, calls, tot, avg, mn, mx in rows[: (60 if '--xyz' in sys.argv else 25)]:
            if md:
                print(f'| `{short(name, 70)}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |')
            else:
                print(f'{short(name, 80):80s} {calls:7d} {tot / 1e6:10.3f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:6.1f}')
    try:
        pmc = cur.execute(
            'select kernel_name, counter_name, grid_size, count(*), avg(value), min(value), max(value) from counters_collection '
            'group by kernel_name, counter_name, grid_size order by avg(value) desc'
        ).fetchall()
    except sqlite3.OperationalError:
        pmc = []
    if pmc:
        print()
        print('| kernel | counter | grid | n | avg | min | max |' if md else f'{"kernel":70s} {"counter":>12s} {"grid":>9s} {"n":>5s} {"avg":>12s} {"min":>12s} {"max":>12s}')
        if md:
            print('|---|---|---:|---:|---:|---:|---:|')
        for name, cname, grid, n, avg, mn, mx in pmc[:20]:
            if md:
                print(f'| `{short(name, 60)}` | {cname} | {grid} | {n} | {avg:.1f} | {mn:.1f} | {mx:.1f} |')
            else:
                print(f'{short(name, 70):70s} {cname:>12s} {grid:9d} {n:5d} {avg:12.1f} {mn:12.1f} {mx:12.1f}')


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Extract the dominant lookup launch's average FETCH_SIZE / WRITE_SIZE (KB per dispatch) from two rocprofv3 --pmc
result databases of the default bench.py run and write profiles-style JSON.
    python tools/pmc_extract.py fetch.db write.db out.json"""
import json
import sqlite3
import sys


def avg(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute(
        "select grid_size, count(*), avg(value), kernel_name from counters_collection where counter_name = ? and kernel_name like '%recency_lookup_%' "
        'group by grid_size, kernel_name order by grid_size desc', (counter,)).fetchall()
    return rows[0]  # the largest grid = the last hop (or the fused hop 0 + hop 1 launch)


g, n, fetch, name = avg(sys.argv[1], 'FETCH_SIZE')
g2, n2, write, _ = avg(sys.argv[2], 'WRITE_SIZE')
assert g == g2
fused = 'fused01' in name
out = {'workload': 'wiki', 'mode': 'ring', 'batch_size': 200, 'num_nbrs': [20, 20], 'slots_per_launch': 12000 * 20 + (600 * 20 if fused else 0),
       'kernel': 'recency_lookup_fused01_kernel (hop 0 + hop 1)' if fused else 'recency_lookup_kernel (hop 1)',
       'grid_threads': g, 'dispatches': n, 'fetch_kb': fetch, 'write_kb': write,
       'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of python bench.py --steps 200 --cpu-batches 0; '
               'KB per dispatch; FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md)'}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out))

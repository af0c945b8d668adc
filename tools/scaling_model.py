#!/usr/bin/env python
"""Modelled multi-GPU scaling from ONE GPU: run the slowest rank's exact step of a W-rank job (bench.py --emulate-world W)
for W = 1, 2, 4, 8.  There is no collective on the sampling data path, so the emulated rank misses nothing; what it cannot
show is interference between processes on a shared host.  Writes profiles/r02_scaling_model.json.

    python tools/scaling_model.py            (on the GPU box, from the repo root)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for workload, scaling, extra in (('wiki', 'weak', []), ('comment', 'weak', ['--steps', '300']), ('comment', 'strong', ['--steps', '300'])):
    rows = []
    for W in (1, 2, 4, 8):
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--scaling', scaling, '--cpu-batches', '0'] + extra
        if W > 1:
            cmd += ['--emulate-world', str(W), '--emulate-rank', str(W - 1)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if not line:
            rows.append({'world': W, 'error': (r.stderr or r.stdout)[-400:]})
            continue
        d = json.loads(line[-1])
        rows.append({'world': W, 'rank_emulated': W - 1, 'ms_per_step': d['ms_per_step'], 'rank_units_per_s': d['value'],
                     'kernel_ms': d['roofline']['avg_kernel_ms'], 'valid_slot_fraction': d['roofline']['valid_slot_fraction']})
    base = rows[0].get('ms_per_step')
    for r in rows:
        if 'ms_per_step' in r and base:
            # weak: per-rank work fixed -> efficiency = t1 / tW, speed-up = W * efficiency; strong: total work fixed -> speed-up = t1 / tW
            r['modelled_speedup'] = (r['world'] * base / r['ms_per_step']) if scaling == 'weak' else base / r['ms_per_step']
            r['modelled_efficiency'] = r['modelled_speedup'] / r['world']
    out[f'{workload}_{scaling}'] = rows
    print(workload, scaling, json.dumps(rows), flush=True)
json.dump({'method': 'bench.py --emulate-world W --emulate-rank W-1 on one MI355X (the last rank; all ranks do the same amount of work); '
           'no hardware multi-GPU curve exists yet (SCALE was skipped)', 'results': out}, open(os.path.join(ROOT, 'profiles', 'r02_scaling_model.json'), 'w'), indent=1)

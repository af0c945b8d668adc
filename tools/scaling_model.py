#!/usr/bin/env python
"""Modelled multi-GPU scaling from ONE GPU: run the slowest rank's exact step of a W-rank job (bench.py --emulate-world W)
for W = 1, 2, 4, 8.  There is no collective on the sampling data path, so the emulated rank misses nothing; what it cannot
show is interference between processes on a shared host.  Writes gpurun_out/profiles_r06/r06_scaling_model.json (copy into profiles/) (the model's inputs -- every emulated run's own numbers -- are the rows of that file).

    python tools/scaling_model.py            (on the GPU box, from the repo root)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
CASES = (('wiki', 'weak', 'ring', []), ('comment', 'weak', 'ring', ['--steps', '200']), ('comment', 'strong', 'ring', ['--steps', '200']),
         ('comment', 'weak', 'csr', ['--steps', '200']), ('comment', 'strong', 'csr', ['--steps', '200']),
         # round 4: the bs=4096 schedule itself dealt round-robin to the ranks (DGDataLoader(batch_shard=)): same batches, same sampled neighbours
         ('comment', 'batch', 'csr', ['--steps', '200']), ('wiki', 'batch', 'csr', []))
for workload, scaling, mode, extra in CASES:
    rows = []
    for W in (1, 2, 4, 8):
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', workload, '--scaling', scaling, '--mode', mode, '--cpu-batches', '0', '--no-default-path', '--extras', 'off'] + extra
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
            # weak / batch: per-rank work fixed (batch: W ranks work through the SAME schedule W batches at a time) -> efficiency = t1 / tW, speed-up = W * efficiency; strong: total work fixed -> speed-up = t1 / tW
            r['modelled_speedup'] = (r['world'] * base / r['ms_per_step']) if scaling in ('weak', 'batch') else base / r['ms_per_step']
            r['modelled_efficiency'] = r['modelled_speedup'] / r['world']
    out[f'{workload}_{scaling}_{mode}'] = rows
    print(workload, scaling, mode, json.dumps(rows), flush=True)
json.dump({'method': 'bench.py --emulate-world W --emulate-rank W-1 on one MI355X (the last rank; all ranks do the same amount of work); '
           'no hardware multi-GPU curve exists yet (SCALE was skipped in every round so far)', 'results': out},
          open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'profiles_r06', 'r06_scaling_model.json'), 'w'), indent=1)

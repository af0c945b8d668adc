#!/usr/bin/env python
"""MFMA-busy fractions from the per-kernel counter table tools/gpu_pmc_cmd.sh prints (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; averages per
dispatch): SQ_* rows are per shader engine (32 of them, 32 SIMDs each), GRBM_GUI_ACTIVE per XCD, so busy fraction of the chip's 1024 SIMDs =
32 x busy / (1024 x cycles) = busy / (32 x cycles).  Whole forward = sum over the forward's kernels weighted by their launches.
    python tools/mfma_json.py in.md out.json"""
import json
import re
import sys

rows = {}
for line in open(sys.argv[1]):
    m = re.match(r'^(.*?)\s+grid=\s*(\d+)\s+(\S+)\s+n=\s*(\d+)\s+avg=([0-9.]+)', line)
    if not m:
        continue
    name, grid, ctr, n, avg = m.group(1).strip(), int(m.group(2)), m.group(3), int(m.group(4)), float(m.group(5))
    rows.setdefault((name, grid), {})[ctr] = (n, avg)
fwd = ('tgat_', 'sgemm_nt', 'gather_leaves', 'gather_compact', 'ln_residual', 'pair_', 'qfold')
busy_all = cyc_all = 0.0
out = {'kernels': []}
tail = None
for (name, grid), c in rows.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
        continue
    (nb, busy), (nc, cyc) = c['SQ_VALU_MFMA_BUSY_CYCLES'], c['GRBM_GUI_ACTIVE']
    launches = nc / 8.0  # GRBM rows: one per XCD
    frac = busy / (32.0 * cyc) if cyc else 0.0
    out['kernels'].append({'kernel': name, 'grid': grid, 'launches': launches, 'mfma_busy_frac': frac, 'gui_active_cycles': cyc})
    if any(k in name for k in fwd):
        busy_all += busy / 32.0 * launches
        cyc_all += cyc * launches
    if 'chain64' in name:
        tail = frac
out['mfma_busy_frac_forward'] = busy_all / cyc_all if cyc_all else None
out['mfma_busy_frac_tail_kernel'] = tail
out['method'] = ('rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python tools/bench_tgat.py 60 by_id (a counter-only pass); per kernel busy / (32 x cycles); '
                 'forward = launch-weighted over the TGAT forward kernels (the sampler kernels excluded)')
import hashlib, os, time
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out['tgat_src_sha'] = hashlib.sha256(open(os.path.join(_root, 'tgm_amd', 'csrc', 'tgat.hip'), 'rb').read()).hexdigest()[:16]  # bench.py labels an older pass as such
out['taken'] = time.strftime('%Y-%m-%d %H:%M:%S UTC', time.gmtime())
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps({k: out[k] for k in ('mfma_busy_frac_forward', 'mfma_busy_frac_tail_kernel')}))

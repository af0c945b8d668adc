#!/bin/bash
# the bench lines that quote the counter profiles, re-taken after the profiles (same kernel sources): driver-args trio, review / comment x 3
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/profiles_r06b
mkdir -p $OUT
b() { o=$1; shift; timeout 600 python bench.py "$@" 2>/dev/null | grep '^{' | tail -1 > $OUT/$o; }
for i in 1 2 3; do b r06_bench_ring_driver_args_$i.json --steps 20 --warmup 5; done
for i in 1 2 3; do b r06_bench_review_ring_$i.json --cpu-batches 0 --workload review; done
for i in 1 2 3; do b r06_bench_comment_ring_$i.json --cpu-batches 0 --workload comment --steps 100; done
for i in 1 2 3; do b r06_bench_comment_csr_$i.json --cpu-batches 0 --workload comment --steps 100 --mode csr; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/profiles_r06b/r06_bench_*_[123].json')):
    d = json.load(open(f)); r = d['roofline']; t = r.get('traffic')
    print(f.split('/')[-1], round(d['ms_per_step'] * 1e3, 2), round(d['value'] / 1e9, 3), round(r['frac'], 3), round(d.get('roofline_hbm_bound', {}).get('frac', 0) if isinstance(d.get('roofline_hbm_bound'), dict) else 0, 3), bool(t and 'bytes' in t))
PY

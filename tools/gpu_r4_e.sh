#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4e
mkdir -p "$OUT"
( time timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_args.out" 2> "$OUT/bench_driver_args.err" ) 2> "$OUT/bench_time.txt"
grep '^{' "$OUT/bench_driver_args.out" | tail -1 > "$OUT/bench_driver_args.json"
cat "$OUT/bench_time.txt"; tail -3 "$OUT/bench_driver_args.err"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4e/bench_driver_args.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'])
r=d['roofline']; print({k:r[k] for k in ('frac','avg_kernel_ms','literal_8d_frac')}); print(json.dumps(r.get('variants'),indent=0)[:1500])
print(json.dumps(d.get('roofline_hbm_bound'),indent=0)[:2500]); print(json.dumps(d.get('aggregation'),indent=0)[:1500])
PY
timeout 600 python tools/find_copies.py > "$OUT/find_copies.txt" 2>&1; tail -45 "$OUT/find_copies.txt"

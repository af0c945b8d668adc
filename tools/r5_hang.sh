#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5_hang
for i in 1 2 3 4 5 6; do
  s=$(date +%s); timeout -k 5 280 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_hang/b$i.json 2>gpurun_out/r5_hang/b$i.err; rc=$?; e=$(date +%s)
  echo "run $i rc=$rc $((e-s))s"
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/r5_hang/b$i.json') if l.startswith('{')][-1])
    print('   hbm_bound s', d.get('roofline_hbm_bound',{}).get('seconds_spent'), 'cfg3 s', d.get('pipeline_cfg3',{}).get('seconds_spent'), 'errors', d.get('extras_errors'), 'cpu', d['cpu_baseline']['sample'][:60])
except Exception as e:
    print('   no json', e)
PY
done

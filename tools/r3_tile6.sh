#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3_tile6
mkdir -p "$OUT"
line() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    print(sys.argv[2], 'ms/step %.4f' % d['ms_per_step'], 'hop1 us %.1f' % (1e3 * r['avg_kernel_ms']), 'frac %.3f' % r['frac'], flush=True)
except Exception as e:
    print(sys.argv[2], 'FAILED', e, flush=True)
PY
}
run() {  # tag, env..., -- args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 150 python bench.py --cpu-batches 0 --no-default-path "$@" 2>"$OUT/$tag.err" | grep '^{' | tail -1 > "$OUT/$tag.json"
  line "$OUT/$tag.json" "$tag"
}
for l in 0 8800 9400 10100 11000 13200; do
  run ring_lds$l TGMX_TILE_LDS=$l -- --workload comment --steps 100
done
run ring_tile0 TGMX_TILE=0 -- --workload comment --steps 100
run ring_noside TGMX_NO_SIDE_STREAM=1 -- --workload comment --steps 100
run ring_tile0_noside TGMX_TILE=0 TGMX_NO_SIDE_STREAM=1 -- --workload comment --steps 100
for l in 0 9400 10100; do
  run csr_lds$l TGMX_TILE_LDS=$l -- --workload comment --steps 100 --mode csr
done

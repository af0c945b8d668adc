#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3_f
mkdir -p "$OUT"
line() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    print(sys.argv[2], 'ms/step %.4f' % d['ms_per_step'], 'hop1 us %.1f' % (1e3 * r['avg_kernel_ms']), 'frac %.3f' % r['frac'], 'G/s %.1f' % (d['value'] / 1e9), flush=True)
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
for i in 1 2; do
run comment_ring X=1 -- --workload comment --steps 200
run comment_csr X=1 -- --workload comment --steps 200 --mode csr
done
run comment_ring_full TGMX_DELTA_WRITES=0 -- --workload comment --steps 200

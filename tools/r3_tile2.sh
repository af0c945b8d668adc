#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3_tile2
mkdir -p "$OUT"
line() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    print(sys.argv[2], 'ms/step %.4f' % d['ms_per_step'], 'hop1 us %.1f' % (1e3 * r['avg_kernel_ms']), 'frac %.3f' % r['frac'], 'MB %.1f' % (r['algorithmic_bytes_per_launch'] / 1e6), flush=True)
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
for ab in 0 16 32 48 64 112; do
  run review_ring_ab$ab TGMX_ABLATE=$ab -- --workload review --steps 300
  run comment_ring_ab$ab TGMX_ABLATE=$ab -- --workload comment --steps 100
done
run review_csr_ab0 TGMX_ABLATE=0 -- --workload review --steps 300 --mode csr
run review_csr_ab48 TGMX_ABLATE=48 -- --workload review --steps 300 --mode csr
run comment_csr_ab0 TGMX_ABLATE=0 -- --workload comment --steps 100 --mode csr
run comment_csr_ab48 TGMX_ABLATE=48 -- --workload comment --steps 100 --mode csr
timeout 300 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_shard_dist_gpu.py -x -q 2>&1 | tail -8

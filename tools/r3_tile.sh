#!/bin/bash
# round 3: the tile kernel (lane-per-seed narrow-row lookup) vs the packed kernel: parity tests + benches
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3_tile
mkdir -p "$OUT"
line() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    print(sys.argv[2], 'ms/step %.4f' % d['ms_per_step'], 'hop1 us %.1f' % (1e3 * r['avg_kernel_ms']), 'frac %.3f' % r['frac'], 'MB %.1f' % (r['algorithmic_bytes_per_launch'] / 1e6), 'default', (d.get('default_path') or {}).get('ms_per_step'), flush=True)
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
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -8
timeout 900 python -m pytest tests/test_sampler_gpu.py -x -q -k "cfg3 or cfg4_comment_shape or fuzz or direct_entry or golden or oracle" 2>&1 | tail -8
for t in 1 0; do
  run comment_ring_tile$t TGMX_TILE=$t -- --workload comment --steps 200
  run comment_ring_full_tile$t TGMX_TILE=$t TGMX_DELTA_WRITES=0 -- --workload comment --steps 200
  run comment_csr_tile$t TGMX_TILE=$t -- --workload comment --steps 200 --mode csr
  run review_ring_tile$t TGMX_TILE=$t -- --workload review --steps 400
  run review_csr_tile$t TGMX_TILE=$t -- --workload review --steps 400 --mode csr
done

#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3_csr
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
run comment_csr X=1 -- --workload comment --steps 200 --mode csr
run review_csr X=1 -- --workload review --steps 300 --mode csr
run wiki_csr X=1 -- --mode csr
timeout 300 python -m pytest tests/test_pipeline_gpu.py tests/test_shard_dist_gpu.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_sampler_gpu.py -x -q -k "csr or cfg4 or static or index or golden" 2>&1 | tail -3

#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4d
mkdir -p "$OUT"
j() { grep '^{' | tail -1; }
timeout 1500 python -m pytest tests/test_tgat_compact_gpu.py tests/test_pipelines_gpu.py tests/test_tgn_gpu.py tests/test_pipeline_gpu.py tests/test_sampler_gpu.py -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/pytest.rc"
tail -4 "$OUT/pytest.log"
for i in 1 2 3; do timeout 300 python tools/bench_tgat.py 200 by_id 2>/dev/null | j | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("tgat by_id", round(d["tgat_forward_us"],1), round(d["sampler_plus_forward_us_per_batch"],1))'; done
timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j | tee "$OUT/bench_tgn.json"
TGMX_BENCH_TGN_NO_LOADER_PASS=1 tools/gpu_trace_byname.sh tgn 300 python $ROOT/tools/bench_tgn.py 200 > "$OUT/tgn_byname.md" 2>&1
cat "$OUT/tgn_byname.md"

#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4j
mkdir -p "$OUT"
j() { grep '^{' | tail -1; }
timeout 1500 python -m pytest tests/test_tgn_gpu.py tests/test_tgn_backward_gpu.py tests/test_tgn_dist_gpu.py tests/test_pipelines_gpu.py -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -4 "$OUT/pytest.log"
for i in 1 2 3; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j | tee -a "$OUT/bench_tgn.jsonl" | cut -c1-60,215-330; done
TGMX_BENCH_TGN_NO_LOADER_PASS=1 tools/gpu_trace_byname.sh tgn 300 python $ROOT/tools/bench_tgn.py 200 > "$OUT/tgn_byname.md" 2>&1
head -24 "$OUT/tgn_byname.md" | cut -c1-170

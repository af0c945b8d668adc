#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4h
mkdir -p "$OUT"
j() { grep '^{' | tail -1; }
timeout 1500 python -m pytest tests/test_tgcn_gpu.py tests/test_pipeline_gpu.py tests/test_tgn_gpu.py tests/test_pipelines_gpu.py -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -8 "$OUT/pytest.log"
for i in 1 2; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j | tee -a "$OUT/bench_tgn.jsonl" | cut -c1-60,215-400; done
TGMX_BENCH_TGN_DENSE=1 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j | tee "$OUT/bench_tgn_dense.json" | cut -c1-60,215-400
TGMX_BENCH_TGN_NO_LOADER_PASS=1 tools/gpu_trace_byname.sh tgn 300 python $ROOT/tools/bench_tgn.py 200 > "$OUT/tgn_byname.md" 2>&1
head -28 "$OUT/tgn_byname.md" | cut -c1-170
tools/gpu_memcopies.sh tgn env TGMX_BENCH_TGN_NO_LOADER_PASS=1 python $ROOT/tools/bench_tgn.py 100 2>&1 | tail -25

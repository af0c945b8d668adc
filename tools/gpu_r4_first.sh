#!/bin/bash
# round 4, first call: is HEAD green on the GPU, and where do the numbers stand
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r4a
mkdir -p "$OUT"
j() { grep '^{' | tail -1; }
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/pytest.rc"
tail -5 "$OUT/pytest.log"
timeout 300 python bench.py --steps 20 --warmup 5 2>"$OUT/bench.err" | j > "$OUT/bench_driver_args.json"
for f in dense by_id; do timeout 300 python tools/bench_tgat.py 200 $f 2>"$OUT/tgat_$f.err" | j > "$OUT/bench_tgat_$f.json"; done
timeout 300 python tools/bench_tgn.py 400 2>"$OUT/tgn.err" | j > "$OUT/bench_tgn.json"
cat "$OUT"/bench_tgat_*.json "$OUT/bench_tgn.json"

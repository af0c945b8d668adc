#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4b
mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/pytest.rc"
tail -5 "$OUT/pytest.log"
for c in 1 0; do
  TGMX_TGAT_COMPACT=$c tools/gpu_trace_cmd.sh tgat_by_id_c$c 30 python $ROOT/tools/bench_tgat.py 60 by_id > "$OUT/tgat_by_id_compact$c.md" 2>&1
done
tools/gpu_trace_cmd.sh tgn 45 python $ROOT/tools/bench_tgn.py 200 > "$OUT/tgn_trace.md" 2>&1
cat "$OUT"/tgat_by_id_compact1.md "$OUT"/tgat_by_id_compact0.md

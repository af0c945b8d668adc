#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4c
mkdir -p "$OUT"
j() { grep '^{' | tail -1; }
timeout 1200 python -m pytest tests/test_tgat_compact_gpu.py tests/test_tgat_gpu.py tests/test_tgat_backward_gpu.py tests/test_pipelines_gpu.py -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee "$OUT/pytest.rc"
tail -4 "$OUT/pytest.log"
run() { tag=$1; shift; env "$@" timeout 300 python tools/bench_tgat.py 200 by_id 2>/dev/null | j > "$OUT/tgat_$tag.json"; echo "$tag $(cat $OUT/tgat_$tag.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["tgat_forward_us"],1), round(d["sampler_plus_forward_us_per_batch"],1))')"; }
run new A=1
run nospan TGMX_ATTN_SPAN=0
run nolds TGMX_PAIR_LDS=0
run nocompact TGMX_TGAT_COMPACT=0
run nocompact_nospan TGMX_TGAT_COMPACT=0 TGMX_ATTN_SPAN=0
tools/gpu_trace_cmd.sh tgat_new 16 python $ROOT/tools/bench_tgat.py 60 by_id > "$OUT/tgat_by_id_new.md" 2>&1
TGMX_TGAT_COMPACT=0 tools/gpu_trace_cmd.sh tgat_nc 14 python $ROOT/tools/bench_tgat.py 60 by_id > "$OUT/tgat_by_id_nocompact.md" 2>&1
cat "$OUT/tgat_by_id_new.md" "$OUT/tgat_by_id_nocompact.md"

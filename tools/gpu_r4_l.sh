#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4l
mkdir -p "$OUT"
j() { grep '^{' | tail -1; }
timeout 1500 python -m pytest tests/test_tgat_gpu.py tests/test_tgat_compact_gpu.py tests/test_tgat_backward_gpu.py tests/test_pipelines_gpu.py -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -6 "$OUT/pytest.log"
run() { tag=$1; shift; env "$@" timeout 300 python tools/bench_tgat.py 200 by_id 2>/dev/null | j > "$OUT/tgat_$tag.json"; echo "$tag $(cat $OUT/tgat_$tag.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["tgat_forward_us"],1), round(d["sampler_plus_forward_us_per_batch"],1))')"; }
run a A=1
run b A=1
run dense_path A=1
tools/gpu_trace_byname.sh tgat_l 170 python $ROOT/tools/bench_tgat.py 60 by_id 2>&1 | head -12 | cut -c1-170

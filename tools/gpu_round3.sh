#!/bin/bash
# Everything the round's profiles/ are made of, in one gpurun call (run from the repo root on the GPU box):
#   tools/gpu_round3.sh   -> gpurun_out/profiles_r03/*   (copy into profiles/ afterwards)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles_r03
mkdir -p "$OUT"
j() { grep '^{' | tail -1; }
b() { out=$1; shift; timeout 300 python bench.py "$@" 2>/dev/null | j > "$OUT/$out"; }
b r03_bench_ring.json
b r03_bench_ring_driver_args.json --steps 20 --warmup 5
b r03_bench_csr.json --cpu-batches 0 --mode csr
b r03_bench_review_ring.json --cpu-batches 0 --workload review
b r03_bench_review_csr.json --cpu-batches 0 --workload review --mode csr
b r03_bench_comment_ring.json --cpu-batches 0 --workload comment --steps 100
b r03_bench_comment_csr.json --cpu-batches 0 --workload comment --steps 100 --mode csr
TGMX_DELTA_WRITES=0 timeout 300 python bench.py --cpu-batches 0 --workload comment --steps 400 2>/dev/null | j > "$OUT/r03_bench_comment_ring_full_writes.json"
TGMX_TILE=0 timeout 300 python bench.py --cpu-batches 0 --workload comment --steps 400 --no-default-path 2>/dev/null | j > "$OUT/r03_bench_comment_ring_packed_kernel.json"
TGMX_TILE=0 timeout 300 python bench.py --cpu-batches 0 --workload comment --steps 400 --mode csr --no-default-path 2>/dev/null | j > "$OUT/r03_bench_comment_csr_packed_kernel.json"
tools/gpu_profile_r3.sh wiki_ring recency_lookup_fused01 20 --warmup 5
tools/gpu_profile_r3.sh comment_csr lookup_tile 100 --workload comment --mode csr
tools/gpu_profile_r3.sh comment_ring lookup_tile 100 --workload comment
for f in dense by_id; do timeout 300 python tools/bench_tgat.py 200 $f 2>/dev/null | j > "$OUT/r03_bench_tgat_$f.json"; done
timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j > "$OUT/r03_bench_tgn.json"
timeout 300 python tools/bench_tgat_train.py 200 2>/dev/null | j > "$OUT/r03_bench_tgat_train.json"
timeout 300 python tools/bench_tgat_train.py 200 by_id 2>/dev/null | j > "$OUT/r03_bench_tgat_train_by_id.json"
TGMX_TGAT_BWD=py timeout 300 python tools/bench_tgat_train.py 200 2>/dev/null | j > "$OUT/r03_bench_tgat_train_composed_backward.json"
tools/gpu_trace_cmd.sh train 40 python $ROOT/tools/bench_tgat_train.py 60 > "$OUT/r03_tgat_train_rocprof_summary.md" 2>/dev/null
rm -f "$OUT/r03_tgat_parity_stats.jsonl"
TGMX_PARITY_STATS="$OUT/r03_tgat_parity_stats.jsonl" timeout 900 python -m pytest tests/test_tgat_gpu.py -q -m gpu -k "reference or headline" > "$OUT/r03_tgat_parity_pytest.log" 2>&1
timeout 1200 python tools/scaling_model.py > "$OUT/scaling_model.log" 2>&1; cp profiles/r03_scaling_model.json "$OUT/" 2>/dev/null
ls -la "$OUT"

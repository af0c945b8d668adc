set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles_r06b
mkdir -p "$OUT"
j() { grep '^{' | tail -1; }
for i in 1 2 3; do timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | j > "$OUT/r06_bench_ring_driver_args_$i.json"; done
for i in 1 2 3; do TGMX_BENCH_TGN_REPEATS=3 TGMX_BENCH_TGN_STREAMS=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j >> "$OUT/r06_bench_tgn_one_stream.jsonl"; done
for i in 1 2 3; do TGMX_BENCH_TGN_REPEATS=3 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j >> "$OUT/r06_bench_tgn.jsonl"; done
TGMX_BENCH_TGN_STREAMS=0 TGMX_BENCH_TGN_NO_LOADER_PASS=1 tools/gpu_trace_byname.sh tgn 300 python $ROOT/tools/bench_tgn.py 200 > "$OUT/r06_tgn_rocprof_summary.md" 2>/dev/null
TGMX_DIST_BACKEND=gloo TGMX_SINGLE_DEVICE=1 TGMX_SCALE_COMMENT_EDGES=4000000 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --cpu-batches 0 2>/dev/null | j > "$OUT/r06_bench_two_ranks_one_gpu.json"
ls -la "$OUT"

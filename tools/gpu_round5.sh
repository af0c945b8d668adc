#!/bin/bash
# Everything round 5's profiles/ are made of (besides tools/gpu_calib_r5.sh: the counter calibration and the stall counters), in one gpurun call (run from the repo root on the GPU box):
#   tools/gpu_round5.sh [quick]  -> gpurun_out/profiles_r05/*   (copy into profiles/ afterwards)
# Not made here (one-off A/Bs of the round's last session, each from a few-line shell loop over the same tools; the commands are in the files' headers
# or in DESIGN 3.3 / 3.5): r05_gemm_lds_ab.txt, r05_gemm_two_row_tiles_rejected.txt (tgmx_sgemm_nt over the training step's / cfg 3's shapes with
# TGMX_GEMM_LDS / a removed TGMX_GEMM_RT knob), r05_tile_w4_rejected.jsonl (bench.py --workload comment with a removed TGMX_TILE_W4 knob),
# r05_bench_tgat_train_by_id_second_box.jsonl and r05_bench_tgn_second_box.jsonl (the same bench commands as below on another box).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles_r05
mkdir -p "$OUT"
QUICK=${1:-}
j() { grep '^{' | tail -1; }
b() { out=$1; shift; timeout 400 python bench.py "$@" 2>/dev/null | j > "$OUT/$out"; }
# the driver's command, three times (min / median of every figure quoted come from these)
for i in 1 2 3; do b r05_bench_ring_driver_args_$i.json --steps 20 --warmup 5; done
b r05_bench_ring.json --extras off
b r05_bench_csr.json --cpu-batches 0 --mode csr --extras off
for i in 1 2 3; do b r05_bench_review_ring_$i.json --cpu-batches 0 --workload review; done
b r05_bench_review_csr.json --cpu-batches 0 --workload review --mode csr
for i in 1 2 3; do b r05_bench_comment_ring_$i.json --cpu-batches 0 --workload comment --steps 100; done
for i in 1 2 3; do b r05_bench_comment_csr_$i.json --cpu-batches 0 --workload comment --steps 100 --mode csr; done
tools/gpu_profile_r5.sh wiki_ring recency_lookup_fused01 20 --warmup 5
tools/gpu_profile_r5.sh comment_csr lookup_tile 100 --workload comment --mode csr
tools/gpu_profile_r5.sh comment_ring lookup_tile 100 --workload comment
tools/gpu_profile_r5.sh review_ring lookup_packed 400 --workload review
# TGAT forward: three repeats per feature mode, a kernel trace and the MFMA counters of the same command
for f in dense by_id; do for i in 1 2 3; do timeout 300 python tools/bench_tgat.py 200 $f 2>/dev/null | j >> "$OUT/r05_bench_tgat_$f.jsonl"; done; done
tools/gpu_trace_byname.sh tgat_fwd 170 python $ROOT/tools/bench_tgat.py 60 by_id > "$OUT/r05_tgat_fwd_rocprof_summary.md" 2>/dev/null
tools/gpu_pmc_cmd.sh tgat_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "" python $ROOT/tools/bench_tgat.py 60 by_id > "$OUT/r05_tgat_mfma_pmc.md" 2>/dev/null
python tools/mfma_json.py "$OUT/r05_tgat_mfma_pmc.md" "$OUT/r05_tgat_mfma_pmc.json"
# cfg 3 (side stream + launch worker: the default of the fast variant; TGMX_BENCH_TGN_STREAMS=0 = one stream, as in round 4)
for i in 1 2 3; do TGMX_BENCH_TGN_STREAMS=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j >> "$OUT/r05_bench_tgn_one_stream.jsonl"; done
TGMX_BENCH_TGN_PHASES=1 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j > "$OUT/r05_bench_tgn_host_phases.json"
for i in 1 2 3; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j >> "$OUT/r05_bench_tgn.jsonl"; done
TGMX_BENCH_TGN_STREAMS=0 TGMX_BENCH_TGN_NO_LOADER_PASS=1 tools/gpu_trace_byname.sh tgn 300 python $ROOT/tools/bench_tgn.py 200 > "$OUT/r05_tgn_rocprof_summary.md" 2>/dev/null
for i in 1 2 3; do timeout 300 python tools/bench_tgcn.py 2>/dev/null | j >> "$OUT/r05_bench_tgcn.jsonl"; done
if [ -z "$QUICK" ]; then
  for i in 1 2 3; do timeout 300 python tools/bench_tgat_train.py 200 by_id 2>/dev/null | j >> "$OUT/r05_bench_tgat_train_by_id.jsonl"; done
  rm -f "$OUT/r05_tgat_parity_stats.jsonl"
  TGMX_PARITY_STATS="$OUT/r05_tgat_parity_stats.jsonl" timeout 900 python -m pytest tests/test_tgat_gpu.py -q -m gpu -k "reference or headline" > "$OUT/r05_tgat_parity_pytest.log" 2>&1
  tools/gpu_trace_byname.sh tgat_train 200 python $ROOT/tools/bench_tgat_train.py 50 by_id > "$OUT/r05_tgat_train_rocprof_summary.md" 2>/dev/null
  # the N > 1 line as the driver would get it, two ranks sharing this one GPU (functional: gloo rendezvous, comment stream shrunk)
  TGMX_DIST_BACKEND=gloo TGMX_SINGLE_DEVICE=1 TGMX_SCALE_COMMENT_EDGES=4000000 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --cpu-batches 0 2>/dev/null | j > "$OUT/r05_bench_two_ranks_one_gpu.json"
  tools/bin/stream_overlap > "$OUT/r05_stream_overlap.jsonl" 2>/dev/null
  python tools/time_gemm_vs_blas.py 2>/dev/null | grep -v amdgpu > "$OUT/r05_gemm_vs_library.txt"
fi
ls -la "$OUT"

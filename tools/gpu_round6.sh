#!/bin/bash
# Everything round 6's profiles/ are made of, in one gpurun call (run from the repo root on the GPU box):
#   tools/gpu_round6.sh [quick]  -> gpurun_out/profiles_r06/*   (copy into profiles/ afterwards)
# Not made here: r06_tile_coop_ab_first_cut.txt / r06_tile_coop_occupancy_sweep.txt (A/B loops of the session that wrote lookup_tile_coop_kernel,
# `bench.py --workload comment` under TGMX_TILE_COOP and two since-removed experiment knobs; the commands are in the files' lines).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles_r06
mkdir -p "$OUT"
QUICK=${1:-}
j() { grep '^{' | tail -1; }
b() { out=$1; shift; timeout 400 python bench.py "$@" 2>/dev/null | j > "$OUT/$out"; }
# the driver's command, three times (min / median of every figure quoted come from these)
for i in 1 2 3; do b r06_bench_ring_driver_args_$i.json --steps 20 --warmup 5; done
b r06_bench_ring.json --extras off
b r06_bench_csr.json --cpu-batches 0 --mode csr --extras off
for i in 1 2 3; do b r06_bench_review_ring_$i.json --cpu-batches 0 --workload review; done
b r06_bench_review_csr.json --cpu-batches 0 --workload review --mode csr
for i in 1 2 3; do b r06_bench_comment_ring_$i.json --cpu-batches 0 --workload comment --steps 100; done
for i in 1 2 3; do b r06_bench_comment_csr_$i.json --cpu-batches 0 --workload comment --steps 100 --mode csr; done
# the same launches through round 5's index phase (one lane per window), same box, same run
for i in 1 2; do TGMX_TILE_COOP=0 b r06_bench_comment_ring_one_lane_per_window_$i.json --cpu-batches 0 --workload comment --steps 100; done
for i in 1 2; do TGMX_TILE_COOP=0 b r06_bench_comment_csr_one_lane_per_window_$i.json --cpu-batches 0 --workload comment --steps 100 --mode csr; done
tools/gpu_profile_r6.sh wiki_ring recency_lookup_fused01 20 --warmup 5
tools/gpu_profile_r6.sh comment_csr lookup_tile 100 --workload comment --mode csr
tools/gpu_profile_r6.sh comment_ring lookup_tile 100 --workload comment
tools/gpu_profile_r6.sh review_ring lookup_packed 400 --workload review
# the share of wave cycles parked in s_waitcnt, before / after: counters only, one pass per variant
for c in 1 0; do
  TGMX_TILE_COOP=$c tools/gpu_pmc_cmd.sh wait_coop$c "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES" lookup_tile python $ROOT/bench.py --cpu-batches 0 --no-default-path --extras off --steps 60 --workload comment > "$OUT/r06_comment_ring_wait_share_coop$c.txt" 2>/dev/null
  TGMX_TILE_COOP=$c tools/gpu_pmc_cmd.sh wait_csr_coop$c "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES" lookup_tile python $ROOT/bench.py --cpu-batches 0 --no-default-path --extras off --steps 60 --workload comment --mode csr > "$OUT/r06_comment_csr_wait_share_coop$c.txt" 2>/dev/null
done
# TGAT forward: three repeats per feature mode, a kernel trace and the MFMA counters of the same command
for f in dense by_id; do for i in 1 2 3; do timeout 300 python tools/bench_tgat.py 200 $f 2>/dev/null | j >> "$OUT/r06_bench_tgat_$f.jsonl"; done; done
tools/gpu_trace_byname.sh tgat_fwd 170 python $ROOT/tools/bench_tgat.py 60 by_id > "$OUT/r06_tgat_fwd_rocprof_summary.md" 2>/dev/null
tools/gpu_pmc_cmd.sh tgat_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "" python $ROOT/tools/bench_tgat.py 60 by_id > "$OUT/r06_tgat_mfma_pmc.md" 2>/dev/null
python tools/mfma_json.py "$OUT/r06_tgat_mfma_pmc.md" "$OUT/r06_tgat_mfma_pmc.json"
# cfg 3 (side stream + launch worker: the default of the fast variant; TGMX_BENCH_TGN_STREAMS=0 = one stream)
for i in 1 2 3; do TGMX_BENCH_TGN_STREAMS=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j >> "$OUT/r06_bench_tgn_one_stream.jsonl"; done
for i in 1 2 3; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j >> "$OUT/r06_bench_tgn.jsonl"; done
TGMX_BENCH_TGN_STREAMS=0 TGMX_BENCH_TGN_NO_LOADER_PASS=1 tools/gpu_trace_byname.sh tgn 300 python $ROOT/tools/bench_tgn.py 200 > "$OUT/r06_tgn_rocprof_summary.md" 2>/dev/null
for i in 1 2 3; do timeout 300 python tools/bench_tgcn.py 2>/dev/null | j >> "$OUT/r06_bench_tgcn.jsonl"; done
if [ -z "$QUICK" ]; then
  for i in 1 2 3; do timeout 300 python tools/bench_tgat_train.py 200 by_id 2>/dev/null | j >> "$OUT/r06_bench_tgat_train_by_id.jsonl"; done
  rm -f "$OUT/r06_tgat_parity_stats.jsonl"
  TGMX_PARITY_STATS="$OUT/r06_tgat_parity_stats.jsonl" timeout 900 python -m pytest tests/test_tgat_gpu.py -q -m gpu -k "reference or headline" > "$OUT/r06_tgat_parity_pytest.log" 2>&1
  tools/gpu_trace_byname.sh tgat_train 200 python $ROOT/tools/bench_tgat_train.py 50 by_id > "$OUT/r06_tgat_train_rocprof_summary.md" 2>/dev/null
  # the N > 1 line as the driver would get it, two ranks sharing this one GPU (functional: gloo rendezvous, comment stream shrunk)
  TGMX_DIST_BACKEND=gloo TGMX_SINGLE_DEVICE=1 TGMX_SCALE_COMMENT_EDGES=4000000 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --cpu-batches 0 2>/dev/null | j > "$OUT/r06_bench_two_ranks_one_gpu.json"
  # the scaling model at this tree (one GPU emulating the last rank of 1 / 2 / 4 / 8)
  timeout 1500 python tools/scaling_model.py "$OUT/r06_scaling_model.json" > "$OUT/r06_scaling_model.log" 2>&1
fi
ls -la "$OUT"

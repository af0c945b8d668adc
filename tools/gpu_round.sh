#!/bin/bash
# Everything the round's profiles/ are made of, in one gpurun call (run from the repo root on the GPU box):
#   tools/gpu_round.sh            -> gpurun_out/profiles_r02/*   (copy into profiles/ afterwards)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/profiles_r02
mkdir -p "$OUT"
j() { grep '^{' | tail -1; }
python bench.py                                              2>/dev/null | j > "$OUT/r02_bench_ring.json"
python bench.py --steps 20 --warmup 5                        2>/dev/null | j > "$OUT/r02_bench_ring_driver_args.json"
TGMX_DELTA_WRITES=0 python bench.py --cpu-batches 0          2>/dev/null | j > "$OUT/r02_bench_ring_full_writes.json"
python bench.py --cpu-batches 0 --pool 0                     2>/dev/null | j > "$OUT/r02_bench_ring_hook_by_hook.json"
python bench.py --cpu-batches 0 --pool 0 --validate sync     2>/dev/null | j > "$OUT/r02_bench_ring_default_sync_validation.json"
python bench.py --cpu-batches 0 --validate sync              2>/dev/null | j > "$OUT/r02_bench_ring_sync_validation_lowered.json"
python bench.py --cpu-batches 0 --pool 4                     2>/dev/null | j > "$OUT/r02_bench_ring_pool4.json"
python bench.py --cpu-batches 0 --mode csr                   2>/dev/null | j > "$OUT/r02_bench_csr.json"
python bench.py --cpu-batches 0 --workload review            2>/dev/null | j > "$OUT/r02_bench_review_ring.json"
python bench.py --cpu-batches 0 --workload comment --steps 400            2>/dev/null | j > "$OUT/r02_bench_comment_ring.json"
python bench.py --cpu-batches 0 --workload comment --steps 400 --mode csr 2>/dev/null | j > "$OUT/r02_bench_comment_csr.json"
python tools/bench_tgat.py        2>/dev/null | j > "$OUT/r02_bench_tgat.json"
python tools/bench_tgat_train.py  2>/dev/null | j > "$OUT/r02_bench_tgat_train.json"
python tools/bench_tgn.py 300 fast       2>/dev/null | j > "$OUT/r02_bench_tgn.json"
python tools/bench_tgn.py 300 reference  2>/dev/null | j > "$OUT/r02_bench_tgn_reference_style_loop.json"
python tools/bench_tgcn.py        2>/dev/null | j > "$OUT/r02_bench_tgcn.json"
rm -f "$OUT/r02_tgat_parity_stats.jsonl"; TGMX_PARITY_STATS="$OUT/r02_tgat_parity_stats.jsonl" python -m pytest tests/test_tgat_gpu.py -q > /dev/null 2>&1
python tools/hbm_probe.py         2>/dev/null > "$OUT/r02_hbm_probe.txt"
python tools/host_time.py --workload wiki   2>/dev/null | tail -1 >  "$OUT/r02_host_vs_device.txt"
python tools/host_time.py --workload review 2>/dev/null | tail -1 >> "$OUT/r02_host_vs_device.txt"
python tools/scaling_model.py > /dev/null 2>&1; cp profiles/r02_scaling_model.json "$OUT/" 2>/dev/null
tools/gpu_trace_cmd.sh tgat 18 python "$ROOT/tools/bench_tgat.py" 50 2>/dev/null | grep -v '^{' > "$OUT/r02_tgat_rocprof_summary.md"
{
  echo '# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python tools/bench_tgat.py 20   (separate counter pass; averages per dispatch;'
  echo '# SQ_* rows are per shader engine (32 of them, 32 SIMDs each), GRBM_GUI_ACTIVE per XCD: MFMA utilisation = 32 x busy / (1024 x cycles) = busy / (32 x cycles))'
  tools/gpu_pmc_cmd.sh mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" tgat_ python "$ROOT/tools/bench_tgat.py" 20
} > "$OUT/r02_tgat_mfma_pmc.md" 2>/dev/null
tools/gpu_profile.sh > "$OUT/gpu_profile.log" 2>&1
cp profiles/r02_sampler_rocprof_summary.md profiles/pmc_hop1.json "$OUT/" 2>/dev/null
rm -rf "$ROOT/gpurun_out/prof_r02"
ls -la "$OUT"

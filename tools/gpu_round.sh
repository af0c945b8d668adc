#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, bench, rocprofv3 kernel trace + PMC passes.
# Usage: tools/gpu_round.sh <tag>     outputs -> gpurun_out/<tag>/  (summaries are then copied to profiles/)
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
python bench.py --mode csr --cpu-batches 0 > $OUT/bench_csr.json 2> $OUT/bench_csr.err; cat $OUT/bench_csr.json
cd /tmp && export TMPDIR=/tmp
# kernel trace + stats (same command as the bench line, fewer steps)
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o trace -- python $ROOT/bench.py --cpu-batches 0 > $OUT/prof_trace.log 2>&1; echo "rocprof trace rc=$?"
# PMC counters: separate passes, no tracing domains (guide: FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- python $ROOT/bench.py --steps 200 --cpu-batches 0 > $OUT/prof_fetch.log 2>&1; echo "rocprof fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o write -- python $ROOT/bench.py --steps 200 --cpu-batches 0 > $OUT/prof_write.log 2>&1; echo "rocprof write rc=$?"
cd $ROOT
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --cpu-batches 0   (ring mode, wiki-shaped, bs=200, k=[20,20])"
  python tools/rocpd_summary.py $OUT/prof_trace/trace_results.db --md
  echo
  echo "# rocprofv3 --pmc FETCH_SIZE -- python bench.py --steps 200 --cpu-batches 0   (KB per dispatch; gfx950: x2 for wide coalesced reads)"
  python tools/rocpd_summary.py $OUT/prof_fetch/fetch_results.db --md | sed -n '/counter/,$p'
  echo
  echo "# rocprofv3 --pmc WRITE_SIZE -- python bench.py --steps 200 --cpu-batches 0   (KB per dispatch)"
  python tools/rocpd_summary.py $OUT/prof_write/write_results.db --md | sed -n '/counter/,$p'
} > $OUT/rocprof_summary.md
python tools/pmc_extract.py $OUT/prof_fetch/fetch_results.db $OUT/prof_write/write_results.db $OUT/pmc_hop1.json
python tools/bench_tgat.py 200 > $OUT/bench_tgat.json 2> $OUT/bench_tgat.err; cat $OUT/bench_tgat.json
python tools/time_update.py > $OUT/time_update.json 2>/dev/null; cat $OUT/time_update.json
python tools/time_step_world.py wiki ring > $OUT/step_by_world.json 2>/dev/null; tail -1 $OUT/step_by_world.json
python tools/bench_tgat_train.py 100 > $OUT/bench_tgat_train.json 2>/dev/null; cat $OUT/bench_tgat_train.json
python tools/bench_tgn.py 300 > $OUT/bench_tgn.json 2>/dev/null; cat $OUT/bench_tgn.json
python tools/bench_tgcn.py > $OUT/bench_tgcn.json 2>/dev/null; cat $OUT/bench_tgcn.json
python bench.py --workload comment --steps 300 --cpu-batches 0 > $OUT/bench_comment_ring.json 2>/dev/null; cut -c1-200 $OUT/bench_comment_ring.json
python bench.py --workload comment --mode csr --steps 300 --cpu-batches 0 > $OUT/bench_comment_csr.json 2>/dev/null; cut -c1-200 $OUT/bench_comment_csr.json
python bench.py --workload review --steps 500 --cpu-batches 0 > $OUT/bench_review_ring.json 2>/dev/null; cut -c1-200 $OUT/bench_review_ring.json
rm -f $OUT/prof_*/*.db   # the raw SQLite traces are tens of MB; the summary is what is kept
head -30 $OUT/rocprof_summary.md

#!/bin/bash
# Profiles of the headline bench command on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats  -> profiles/r02_sampler_rocprof_summary.md (+ launch-gap summary)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (never combined with a trace) -> profiles/pmc_hop1.json
# Usage: tools/gpu_profile.sh [extra bench.py args]
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p "$OUT" profiles
export TMPDIR=/tmp
ARGS="--cpu-batches 0 $*"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" $ARGS) > "$OUT/trace.log" 2>&1
grep '^{' "$OUT/trace.log" | tail -1 > "$OUT/bench_under_trace.json"
{
  python tools/prof_summary.py trace "$OUT/trace" --title "rocprofv3 --kernel-trace --stats -- python bench.py $ARGS   (ring mode, wiki-shaped, bs=200, k=[20,20])"
  echo
  echo "# the dominant kernel over bench.py's TIMED steps only (the last 393 launches; the table above also averages over the untimed ring fill)"
  echo '| what | avg | min | max |'
  echo '|---|---|---|---|'
  python tools/prof_summary.py tail "$OUT/trace" --kernel recency_lookup_fused01_kernel --last 393
  echo
  echo '# launch-to-launch accounting on the stream, steady part of the run (tools/prof_summary.py gaps)'
  echo '```'
  python tools/prof_summary.py gaps "$OUT/trace" --kernel recency_lookup_fused01
  echo '```'
} > profiles/r02_sampler_rocprof_summary.md
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $C -d "$OUT/pmc_$C" -- python "$ROOT/bench.py" $ARGS --steps 200) > "$OUT/pmc_$C.log" 2>&1
  {
    echo
    echo "# rocprofv3 --pmc $C -- python bench.py $ARGS --steps 200   (KB per dispatch; gfx950: FETCH_SIZE x2 for wide coalesced reads)"
    python tools/prof_summary.py pmc "$OUT/pmc_$C" | head -8
  } >> profiles/r02_sampler_rocprof_summary.md
done
python - "$OUT" $ARGS <<'PY'
import json, sqlite3, glob, sys, os
out = sys.argv[1]
def avg(counter):
    db = sqlite3.connect(sorted(glob.glob(os.path.join(out, f'pmc_{counter}', '**', '*.db'), recursive=True))[-1])
    r = db.execute("select k.grid_x, count(*), avg(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
                   "where k.name like '%recency_lookup_fused01%' and p.counter_name = ? group by k.grid_x order by count(*) desc", (counter,)).fetchone()
    return r
f, w = avg('FETCH_SIZE'), avg('WRITE_SIZE')
json.dump({'workload': 'wiki', 'mode': 'ring', 'batch_size': 200, 'num_nbrs': [20, 20], 'slots_per_launch': 252000,
           'kernel': 'recency_lookup_fused01_kernel (hop 0 + hop 1)', 'grid_threads': f[0], 'dispatches': f[1], 'fetch_kb': f[2], 'write_kb': w[2],
           'bench_args': ' '.join(sys.argv[2:]),
           'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of python bench.py <bench_args> --steps 200; KB per dispatch; '
                   'FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md)'}, open('profiles/pmc_hop1.json', 'w'), indent=1)
PY
cat profiles/r02_sampler_rocprof_summary.md | head -40
cat "$OUT/bench_under_trace.json"

#!/bin/bash
# Round-6 profiles of a bench.py command on the GPU box: kernel trace + FETCH_SIZE / WRITE_SIZE counters, both restricted to the
# launches bench.py TIMES (the last <steps> dispatches of the dominant kernel: --no-default-path keeps them last).
#   tools/gpu_profile_r5.sh <tag> <kernel substring> <steps> [bench.py args ...]
# writes gpurun_out/profiles_r06/r06_<tag>_rocprof_summary.md and r06_<tag>_pmc.json (counters in separate passes, never with a trace);
# copy them into profiles/ afterwards (only gpurun_out/ travels back from the GPU box)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=$1; KERN=$2; STEPS=$3; shift 3
OUT=$ROOT/gpurun_out/prof_r06_$TAG
PROF=$ROOT/gpurun_out/profiles_r06
rm -rf "$OUT"; mkdir -p "$OUT" "$PROF"
export TMPDIR=/tmp
ARGS="--cpu-batches 0 --no-default-path --extras off --steps $STEPS $*"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" $ARGS) > "$OUT/trace.log" 2>&1
grep '^{' "$OUT/trace.log" | tail -1 > "$OUT/bench_under_trace.json"
{
  python tools/prof_summary.py trace "$OUT/trace" --title "rocprofv3 --kernel-trace --stats -- python bench.py $ARGS" | head -14
  echo
  echo "# the dominant kernel over bench.py's TIMED steps only (its last $STEPS launches; the table above also averages the untimed ring fill and the warm-up)"
  echo '| what | avg | min | max |'
  echo '|---|---|---|---|'
  python tools/prof_summary.py tail "$OUT/trace" --kernel "$KERN" --last "$STEPS"
} > "$PROF/r06_${TAG}_rocprof_summary.md"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $C -d "$OUT/pmc_$C" -- python "$ROOT/bench.py" $ARGS) > "$OUT/pmc_$C.log" 2>&1
  python tools/prof_summary.py pmctail "$OUT/pmc_$C" --kernel "$KERN" --last "$STEPS" > "$OUT/pmc_$C.json" 2>>"$OUT/pmc_$C.log" || echo '{}' > "$OUT/pmc_$C.json"
done
python - "$OUT" "$TAG" "$KERN" "$STEPS" "$ARGS" "$PROF" <<'PY'
import json, sys
out, tag, kern, steps, args, prof = sys.argv[1:7]
f = json.load(open(f'{out}/pmc_FETCH_SIZE.json')); w = json.load(open(f'{out}/pmc_WRITE_SIZE.json'))
bench = {}
try:
    bench = json.loads(open(f'{out}/bench_under_trace.json').read())
except Exception:
    pass
res = {'bench_args': args, 'kernel': kern, 'dispatches_counted': f.get('dispatches'), 'which': f'the last {steps} dispatches = the timed steps',
       'fetch_kb': f.get('counters', {}).get('FETCH_SIZE', {}).get('avg'), 'write_kb': w.get('counters', {}).get('WRITE_SIZE', {}).get('avg'),
       'grid_x': f.get('grid_x'), 'slots_per_launch': None,
       'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of exactly this command; KB per dispatch; FETCH_SIZE must be doubled on '
               'gfx950 (MI355X_MICROARCH.md: 128-byte requests tallied at 64 B)'}
r = bench.get('roofline') or {}
res['algorithmic_bytes_per_launch_under_trace'] = r.get('algorithmic_bytes_per_launch')
res['profile_key'] = (bench.get('config') or {}).get('profile_key')
import hashlib, os, time
h = hashlib.sha256()
for name in ('recency.hip', 'pipeline.hip', 'common.h'):
    h.update(open(os.path.join('tgm_amd', 'csrc', name), 'rb').read())
res['kernel_src_sha'] = h.hexdigest()[:16]  # bench.py quotes these counters only while the tree's kernel sources are these
res['taken'] = time.strftime('%Y-%m-%d %H:%M:%S UTC', time.gmtime())
json.dump(res, open(f'{prof}/r06_{tag}_pmc.json', 'w'), indent=1)
print(json.dumps(res))
PY
cat "$PROF/r06_${TAG}_rocprof_summary.md" | tail -6
rm -rf "$OUT/trace" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE

#!/bin/bash
# rocprofv3 --kernel-trace of any bench command; prints the per-kernel table.  Usage: tools/gpu_trace.sh <tag> <bench args...>
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=$1; shift
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT" -- python "$ROOT/bench.py" --cpu-batches 0 "$@") > "$OUT.log" 2>&1
grep '^{' "$OUT.log" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
python tools/prof_summary.py trace "$OUT" --title "rocprofv3 --kernel-trace --stats -- python bench.py --cpu-batches 0 $*" | head -${ROWS:-18}
python tools/prof_summary.py gaps "$OUT" --kernel "${GAPK:-lookup}"

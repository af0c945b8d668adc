#!/bin/bash
# rocprofv3 --kernel-trace of an arbitrary python command; prints the per-kernel-NAME table.  Usage: tools/gpu_trace_byname.sh <tag> <steps> python ...
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=$1; STEPS=$2; shift 2
OUT=$ROOT/gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT" -- "$@") > "$OUT.log" 2>&1
grep '^{' "$OUT.log" | tail -1
python tools/prof_summary.py byname "$OUT" --steps "$STEPS" --title "rocprofv3 --kernel-trace --stats -- $*"
rm -rf "$OUT"

#!/bin/bash
# kernel timeline of two steady-state sampler steps (rocprofv3 --kernel-trace; queue id = stream): tools/gpu_sampler_timeline.sh <anchor kernel substring> [bench.py args ...]
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
ANCHOR=$1; shift
OUT=$ROOT/gpurun_out/sampler_timeline
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT" -- python "$ROOT/bench.py" --cpu-batches 0 --no-default-path --extras off "$@") > "$OUT.log" 2>&1
grep '^{' "$OUT.log" | tail -1 | cut -c 1-300
python tools/step_timeline.py "$OUT" "$ANCHOR"
rm -rf "$OUT"

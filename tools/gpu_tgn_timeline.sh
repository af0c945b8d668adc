#!/bin/bash
# kernel timeline of a steady-state cfg 3 batch (rocprofv3 --kernel-trace; queue id = stream): tools/gpu_tgn_timeline.sh [env assignments ...]
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/tgn_timeline
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp TGMX_BENCH_TGN_NO_LOADER_PASS=1
for kv in "$@"; do export "$kv"; done
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT" -- python "$ROOT/tools/bench_tgn.py" 200) > "$OUT.log" 2>&1
grep '^{' "$OUT.log" | tail -1 | cut -c 200-520
python tools/step_timeline.py "$OUT" tgn_aggregate_kernel

#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
for t in 1 0; do
  OUT=$ROOT/gpurun_out/tl_$t
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && TGMX_TILE=$t timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT -- python $ROOT/bench.py --cpu-batches 0 --no-default-path --workload comment --steps 20 --warmup 4) > $OUT.log 2>&1
  echo "=== TGMX_TILE=$t"
  python tools/step_timeline.py $OUT recency_lookup_kernel
  rm -rf $OUT
done

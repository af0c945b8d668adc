#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r3_tile4
mkdir -p $OUT
TGMX_TILE_DBG=$PWD/$OUT/ring.bin timeout 200 python bench.py --cpu-batches 0 --no-default-path --workload comment --steps 12 --warmup 2 > /dev/null 2>&1
python tools/tile_dbg.py $OUT/ring.bin
TGMX_TILE_DBG=$PWD/$OUT/csr.bin timeout 200 python bench.py --cpu-batches 0 --no-default-path --workload comment --steps 12 --warmup 2 --mode csr > /dev/null 2>&1
python tools/tile_dbg.py $OUT/csr.bin
TGMX_TILE_DBG=$PWD/$OUT/review.bin timeout 200 python bench.py --cpu-batches 0 --no-default-path --workload review --steps 12 --warmup 2 > /dev/null 2>&1
python tools/tile_dbg.py $OUT/review.bin
rm -f $OUT/*.bin

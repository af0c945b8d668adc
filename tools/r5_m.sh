#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_tgn_gpu.py -x -q -m gpu -k "tgn_step" 2>&1 | tail -5
for i in 1 2 3; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 | cut -c200-420; done
echo "-- three module calls"; for i in 1 2; do TGMX_BENCH_TGN_STEP=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 | cut -c200-420; done
echo "-- one stream"; TGMX_BENCH_TGN_STREAMS=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 | cut -c200-420

#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_tgn_gpu.py tests/test_pipelines_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 | cut -c200-420; done
echo "-- no overlap inside the step"; for i in 1 2; do TGMX_TGN_STEP_OVERLAP=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 | cut -c200-420; done
echo "-- one stream loader, overlap inside the step"; TGMX_BENCH_TGN_STREAMS=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 | cut -c200-420

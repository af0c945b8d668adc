#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_tgn_gpu.py tests/test_pipelines_gpu.py -x -q -m gpu 2>&1 | tail -3
TGMX_BENCH_TGN_STREAMS=0 TGMX_BENCH_TGN_NO_LOADER_PASS=1 tools/gpu_trace_byname.sh tgn 300 python $PWD/tools/bench_tgn.py 200 2>/dev/null | grep -E "group_|launches"
for i in 1 2 3; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 | cut -c200-330; done
for i in 1 2; do TGMX_TCONV_SCAN_PLACE=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 | cut -c200-330; done

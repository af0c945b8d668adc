#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_tgat_gpu.py -x -q 2>&1 | tail -6
timeout 300 python tools/bench_tgat.py 200 dense 2>/dev/null | tail -1
timeout 300 python tools/bench_tgat.py 200 by_id 2>/dev/null | tail -1

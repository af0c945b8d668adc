#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_tgat_backward_gpu.py -x -q -m gpu 2>&1 | tail -2
for w in 1; do echo "WPB=$w"; TGMX_ATTN_BWD_WPB=$w tools/gpu_trace_byname.sh tgat_train_$w 200 python $PWD/tools/bench_tgat_train.py 50 by_id 2>/dev/null | grep "attn_backward" | cut -c1-170; done
timeout 300 python tools/bench_tgat_train.py 200 by_id 2>/dev/null | grep '^{' | tail -1 | cut -c200-330

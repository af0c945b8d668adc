#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_g; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_tgat_backward_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 300 python tools/bench_tgat_train.py 200 by_id 2>/dev/null | grep '^{' | tail -1 >> $O/bench_tgat_train_by_id.jsonl
tools/gpu_trace_byname.sh tgat_train 200 python $PWD/tools/bench_tgat_train.py 50 by_id 2>/dev/null | head -9 | cut -c1-180
tail -3 $O/pytest.log; cut -c150-420 $O/bench_tgat_train_by_id.jsonl

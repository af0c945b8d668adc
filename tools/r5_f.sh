#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_f; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_pipelines_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for i in 1 2 3; do timeout 300 python tools/bench_tgn.py 400 2>$O/err.log | grep '^{' | tail -1 >> $O/bench_tgn_side.jsonl; done
TGMX_LOADER_WORKER=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep "^{" | tail -1 >> $O/bench_tgn_noworker.jsonl
TGMX_BENCH_TGN_STREAMS=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 >> $O/bench_tgn_one.jsonl
TGMX_BENCH_TGN_PHASES=1 python tools/bench_tgn.py 400 2>/dev/null | grep "^{" > $O/phases.json
tail -4 $O/pytest.log; cut -c200-560 $O/bench_tgn_side.jsonl $O/bench_tgn_noworker.jsonl $O/bench_tgn_one.jsonl; cat $O/phases.json; tail -3 $O/err.log

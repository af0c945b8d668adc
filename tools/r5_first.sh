#!/bin/bash
# round 5, first call: the new multi-rank tests + where the round starts (TGAT forward, cfg 3, headline line)
cd "$(dirname "$0")/.."
O=gpurun_out/r5_first; mkdir -p $O
timeout 1200 python -m pytest tests/test_shard_dist_gpu.py tests/test_bench_launcher.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "pytest rc $?" >> $O/pytest_new.log
timeout 300 python bench.py --steps 20 --warmup 5 2>$O/bench.err | grep '^{' | tail -1 > $O/bench_driver_args.json
for i in 1 2; do timeout 300 python tools/bench_tgat.py 200 by_id 2>/dev/null | grep '^{' | tail -1 >> $O/bench_tgat_by_id.jsonl; done
for i in 1 2; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 >> $O/bench_tgn.jsonl; done
tail -5 $O/pytest_new.log; cat $O/bench_tgat_by_id.jsonl | cut -c1-600; cat $O/bench_tgn.jsonl | cut -c1-600

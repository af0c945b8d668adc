#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_d; mkdir -p $O
for i in 1 2; do TGMX_BENCH_TGN_STREAMS=1 timeout 300 python tools/bench_tgn.py 400 2>$O/err_$i.log | grep '^{' | tail -1 >> $O/bench_tgn_streams.jsonl; done
timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 >> $O/bench_tgn_one.jsonl
cut -c200-620 $O/bench_tgn_streams.jsonl $O/bench_tgn_one.jsonl; tail -3 $O/err_1.log

#!/bin/bash
# interleaved A/B of cfg 3 (tools/bench_tgn.py 400) over environment settings: tools/gpu_cfg3_ab_r6.sh ROUNDS "A=1 B=2" "C=3" ...
ROUNDS=$1; shift
run() { env $1 python tools/bench_tgn.py 400 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-60s %.1f busy %.1f wait %.1f' % (sys.argv[1], d['pipeline_us_per_batch'], d['host_busy_us_per_batch'], d['host_waiting_for_the_device_us_per_batch']))" "$1"; }
for i in $(seq $ROUNDS); do for cfg in "$@"; do run "$cfg"; done; done

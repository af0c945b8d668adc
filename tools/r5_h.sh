#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_tgat_backward_gpu.py -x -q -m gpu 2>&1 | tail -2
tools/gpu_trace_byname.sh tgat_train 200 python $PWD/tools/bench_tgat_train.py 50 by_id 2>/dev/null | grep -E "attn_backward|launches" | cut -c1-170
for i in 1 2; do timeout 300 python tools/bench_tgat_train.py 200 by_id 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['train_step_us_fixed_batch'],1), round(d['sampler_plus_train_step_us_per_batch'],1), round(d['adam_fused']['train_step_us_fixed_batch'],1), round(d['adam_fused']['sampler_plus_train_step_us_per_batch'],1))"; done

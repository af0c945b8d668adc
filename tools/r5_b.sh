#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_b; mkdir -p $O
timeout 1500 python -m pytest tests/test_tgn_gpu.py tests/test_pipelines_gpu.py tests/test_gemm_gpu.py tests/test_tgn_backward_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for i in 1 2; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 >> $O/bench_tgn.jsonl; done
TGMX_GEMM_PAIR=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 >> $O/bench_tgn_nopair.jsonl
timeout 300 python bench.py --steps 20 --warmup 5 --extras off 2>$O/bench.err | grep '^{' | tail -1 > $O/bench_driver_args.json
TGMX_BENCH_TGN_NO_LOADER_PASS=1 tools/gpu_trace_byname.sh tgn 300 python $PWD/tools/bench_tgn.py 200 > $O/tgn_rocprof_summary.md 2>/dev/null
tail -4 $O/pytest.log; cut -c1-330 $O/bench_tgn.jsonl $O/bench_tgn_nopair.jsonl; head -25 $O/tgn_rocprof_summary.md
python -c "
import json;d=json.load(open('$O/bench_driver_args.json'));print(d['ms_per_step'],d['default_path'])"

#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_c; mkdir -p $O
timeout 1500 python -m pytest tests/test_tgn_gpu.py tests/test_pipelines_gpu.py tests/test_gemm_gpu.py tests/test_tgn_backward_gpu.py "tests/test_sampler_gpu.py::test_cfg3_review_full_size_midstream_properties" -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for i in 1 2; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 >> $O/bench_tgn.jsonl; done
TGMX_TCONV_RIDE=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | grep '^{' | tail -1 >> $O/bench_tgn_noride.jsonl
TGMX_BENCH_TGN_NO_LOADER_PASS=1 tools/gpu_trace_byname.sh tgn 300 python $PWD/tools/bench_tgn.py 200 > $O/tgn_rocprof_summary.md 2>/dev/null
tail -4 $O/pytest.log; cut -c1-330 $O/bench_tgn.jsonl $O/bench_tgn_noride.jsonl; head -25 $O/tgn_rocprof_summary.md

#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_tgn_gpu.py tests/test_tgn_backward_gpu.py tests/test_tgn_dist_gpu.py tests/test_pipelines_gpu.py -x -q 2>&1 | tail -4
for args in "--workload comment --scaling strong --mode csr --emulate-world 4 --emulate-rank 3" "--workload comment --scaling strong --mode csr --emulate-world 8 --emulate-rank 7" "--workload comment --scaling strong --mode csr --emulate-world 2 --emulate-rank 1" "--workload review"; do
  timeout 200 python bench.py --cpu-batches 0 --no-default-path --steps 200 $args 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args', 'ms/step %.4f' % d['ms_per_step'], 'kernel us %.1f' % (1e3*d['roofline']['avg_kernel_ms']))"
done
timeout 300 python tools/bench_tgn.py 300 fast 2>/dev/null | tail -1
timeout 600 python tools/profile_tgn_host.py 2>/dev/null | head -70

#!/bin/bash
# the round's last measurement pass on one box: sampler / pipeline tests, the four counter + trace profiles of the final kernel sources, the driver-args trio
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/profiles_r06 gpurun_out/profiles_r06b
python -m pytest tests/test_sampler_gpu.py tests/test_pipelines_gpu.py tests/test_tgn_gpu.py -m gpu -q 2>&1 | tail -2
tools/gpu_profile_r6.sh wiki_ring recency_lookup_fused01 20 --warmup 5 > /dev/null 2>&1
tools/gpu_profile_r6.sh comment_csr lookup_tile 100 --workload comment --mode csr > /dev/null 2>&1
tools/gpu_profile_r6.sh comment_ring lookup_tile 100 --workload comment > /dev/null 2>&1
tools/gpu_profile_r6.sh review_ring lookup_packed 400 --workload review > /dev/null 2>&1
ls -la gpurun_out/profiles_r06
for i in 1 2 3; do timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > gpurun_out/profiles_r06b/r06_bench_ring_driver_args_$i.json; done
python - <<'PY'
import json
for i in (1, 2, 3):
    d = json.load(open(f'gpurun_out/profiles_r06b/r06_bench_ring_driver_args_{i}.json'))
    print(i, round(d['ms_per_step'] * 1e3, 2), round(d['value'] / 1e9, 3), round(d['roofline']['frac'], 3), d['roofline']['traffic'] is not None and 'bytes' in d['roofline']['traffic'])
PY

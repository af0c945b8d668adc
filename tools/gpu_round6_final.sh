#!/bin/bash
# the round's last measurement pass on one box: sampler / pipeline tests, the four counter + trace profiles of the final kernel sources, the driver-args trio
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/profiles_r06 gpurun_out/profiles_r06b
python -m pytest tests -m gpu -q 2>&1 | tail -2
tools/gpu_profile_r6.sh wiki_ring recency_lookup_fused01 20 --warmup 5 > /dev/null 2>&1
tools/gpu_profile_r6.sh comment_csr lookup_tile 100 --workload comment --mode csr > /dev/null 2>&1
tools/gpu_profile_r6.sh comment_ring lookup_tile 100 --workload comment > /dev/null 2>&1
tools/gpu_profile_r6.sh review_ring lookup_packed 400 --workload review > /dev/null 2>&1
ls -la gpurun_out/profiles_r06
# the bench lines quote the counter profiles committed under profiles/: put the fresh ones there first (on this box; copy gpurun_out/profiles_r06* home afterwards)
cp gpurun_out/profiles_r06/* profiles/
tools/gpu_round6_lines.sh

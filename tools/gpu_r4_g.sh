#!/bin/bash
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4g
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_tgcn_gpu.py tests/test_sampler_gpu.py -m gpu -q -x -k "tgcn or gcn or riders or dirty" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -15 "$OUT/pytest.log"
export TMPDIR=/tmp
(cd /tmp && timeout 120 rocprofv3 -L) > "$OUT/counters_all.txt" 2>&1
grep -i -E 'hbm|umc|mall|dram|EA_|MC_|fabric|TCC_EA|TCC_BUBBLE|RDREQ|WRREQ' "$OUT/counters_all.txt" | cut -c1-220 | sort -u | head -80 > "$OUT/counters_dram_side.txt"
wc -l "$OUT/counters_all.txt"; cat "$OUT/counters_dram_side.txt" | head -70
tools/gpu_memcopies.sh tgn env TGMX_BENCH_TGN_NO_LOADER_PASS=1 python $ROOT/tools/bench_tgn.py 100 2>&1 | tail -40

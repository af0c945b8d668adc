#!/bin/bash
# round 3, call 2: ablations of the packed narrow-row lookup (comment-shaped hop 1) + PMC passes on the stateless (csr) variant
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3_diag2
mkdir -p "$OUT"
line() {
python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    print(sys.argv[2], 'ms/step %.4f' % d['ms_per_step'], 'hop1 us %.1f' % (1e3 * r['avg_kernel_ms']), 'frac %.3f' % r['frac'], 'MB %.1f' % (r['algorithmic_bytes_per_launch'] / 1e6), flush=True)
except Exception as e:
    print(sys.argv[2], 'FAILED', e, flush=True)
PY
}
run() {  # tag, env..., -- args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 120 python bench.py --cpu-batches 0 --no-default-path --workload comment --steps 120 "$@" 2>"$OUT/$tag.err" | grep '^{' | tail -1 > "$OUT/$tag.json"
  line "$OUT/$tag.json" "$tag"
}
for ab in 0 1 2 3 4 8 11 15; do run ring_ab$ab TGMX_ABLATE=$ab --; done
for ab in 0 1 2 3 4 15; do run ringfull_ab$ab TGMX_ABLATE=$ab TGMX_DELTA_WRITES=0 --; done
for ab in 0 1 3 4; do run csr_ab$ab TGMX_ABLATE=$ab -- --mode csr; done
for nb in 1792 3584 7168 14336; do run ring_blocks$nb TGMX_PACKED_BLOCKS=$nb --; done
for nb in 1792 3584 7168; do run csr_blocks$nb TGMX_PACKED_BLOCKS=$nb -- --mode csr; done
# PMC on the stateless variant (no ring update: the large-batch update path crashed under rocprofv3 --pmc in call 1)
P="python $ROOT/bench.py --cpu-batches 0 --no-default-path --workload comment --mode csr --steps 24 --warmup 4"
K=lookup_packed
pmc() { tag=$1; shift; timeout 150 tools/gpu_pmc_cmd.sh $tag "$1" $K $P > "$OUT/pmc_$tag.txt" 2>&1; tail -n 12 "$OUT/pmc_$tag.txt"; }
pmc a "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"
pmc b "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
pmc c "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum"
pmc d "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"
pmc f "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
pmc g "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -15
timeout 300 python -m pytest tests/test_sampler_gpu.py -x -q -k "cfg2_benched" 2>&1 | tail -5
timeout 120 python bench.py --steps 20 --warmup 5 2>&1 | grep '^{' | tail -1 > "$OUT/bench_default.json"; python -c "
import json; d=json.load(open('$OUT/bench_default.json')); print('headline', d['ms_per_step'], d['value'], 'default', d.get('default_path'))"

#!/bin/bash
# round 3, call 1: where does the narrow-row (comment-shaped) hop-1 lookup spend its time?  baselines + PMC passes.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3_diag1
mkdir -p "$OUT"
j() { grep '^{' | tail -1; }
T0=$(date +%s)
python bench.py --cpu-batches 0 --workload comment --steps 200 2>/dev/null | j > "$OUT/comment_ring.json"
echo "comment ring bench: $(( $(date +%s) - T0 )) s"; T0=$(date +%s)
python bench.py --cpu-batches 0 --workload comment --steps 200 --mode csr 2>/dev/null | j > "$OUT/comment_csr.json"
echo "comment csr bench: $(( $(date +%s) - T0 )) s"
TGMX_DELTA_WRITES=0 python bench.py --cpu-batches 0 --workload comment --steps 200 2>/dev/null | j > "$OUT/comment_ring_full.json"
python bench.py --cpu-batches 0 --workload review --steps 400 2>/dev/null | j > "$OUT/review_ring.json"
for f in comment_ring comment_csr comment_ring_full review_ring; do
  python - "$OUT/$f.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d['roofline']
print(sys.argv[1].split('/')[-1], 'ms/step %.4f' % d['ms_per_step'], 'hop1 us %.1f' % (1e3 * r['avg_kernel_ms']), 'frac %.3f' % r['frac'], 'MB %.1f' % (r['algorithmic_bytes_per_launch'] / 1e6), 'valid %.3f' % r['valid_slot_fraction'])
PY
done
P="python $ROOT/bench.py --cpu-batches 0 --workload comment --steps 40 --warmup 4"
K=lookup_packed
tools/gpu_pmc_cmd.sh a "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" $K $P > "$OUT/pmc_a.txt" 2>&1
tools/gpu_pmc_cmd.sh b "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" $K $P > "$OUT/pmc_b.txt" 2>&1
tools/gpu_pmc_cmd.sh c "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" $K $P > "$OUT/pmc_c.txt" 2>&1
tools/gpu_pmc_cmd.sh d "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" $K $P > "$OUT/pmc_d.txt" 2>&1
tools/gpu_pmc_cmd.sh e "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" $K $P > "$OUT/pmc_e.txt" 2>&1
tools/gpu_pmc_cmd.sh f "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" $K $P > "$OUT/pmc_f.txt" 2>&1
tools/gpu_pmc_cmd.sh g "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" $K $P > "$OUT/pmc_g.txt" 2>&1
tools/gpu_pmc_cmd.sh h "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum" $K $P > "$OUT/pmc_h.txt" 2>&1
tools/gpu_pmc_cmd.sh i "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_IB_STALL_sum GRBM_UTCL2_BUSY" $K $P > "$OUT/pmc_i.txt" 2>&1
tail -n 30 "$OUT"/pmc_*.txt

#!/bin/bash
# instruction-fetch / scalar-cache side of the attention forward (counter passes only): is the 118 KB straight-line kernel fetch-bound?
cd "$(dirname "$0")/.."
O=gpurun_out/r5_ifetch; rm -rf $O; mkdir -p $O
{
for C in "SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_REQ SQC_ICACHE_HITS" "SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB" "SQC_DCACHE_REQ SQC_DCACHE_MISSES" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tools/gpu_pmc_cmd.sh tgat_attn_if "$C" "attn_reduce_reg" python $PWD/tools/bench_tgat.py 60 by_id 2>/dev/null
done
echo "# TGMX_ATTN_SPAN=0 (every row through the all-slots body: one code path)"
for C in "SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  TGMX_ATTN_SPAN=0 tools/gpu_pmc_cmd.sh tgat_attn_if "$C" "attn_reduce_reg" python $PWD/tools/bench_tgat.py 60 by_id 2>/dev/null
done
} > $O/ifetch.txt 2>&1
cut -c1-150 $O/ifetch.txt

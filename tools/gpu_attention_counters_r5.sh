#!/bin/bash
# where the attention kernels' wave-cycles go (forward: inference, 12 600 + 600 rows; backward: training step) -- counter passes only
cd "$(dirname "$0")/.."
O=gpurun_out/r5_l; rm -rf $O; mkdir -p $O
{
echo "# rocprofv3 --pmc over tools/bench_tgat.py 60 by_id (forward) and tools/bench_tgat_train.py 30 by_id (backward); per-dispatch sums over all shader engines, averaged per kernel and grid"
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVES"; do
  tools/gpu_pmc_cmd.sh tgat_attn_f "$C" "attn_reduce" python $PWD/tools/bench_tgat.py 60 by_id 2>/dev/null
  tools/gpu_pmc_cmd.sh tgat_attn_b "$C" "attn_backward" python $PWD/tools/bench_tgat_train.py 30 by_id 2>/dev/null
done
} > $O/attention_counters.txt 2>&1
cat $O/attention_counters.txt | cut -c1-160

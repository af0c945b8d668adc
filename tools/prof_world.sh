#!/bin/bash
# rocprofv3 kernel trace of rank 0's step of a W-rank job (tools/time_step_world.py), riders on and off.
# Usage: tools/prof_world.sh [workload] [worlds...]    outputs -> gpurun_out/prof_world/
WL=${1:-wiki}; shift
WORLDS=${@:-1 8}
ROOT=$PWD; OUT=$ROOT/gpurun_out/prof_world; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in $WORLDS; do
  for tag in ride noride; do
    if [ $tag = noride ]; then export TGMX_NO_RIDE=1; else unset TGMX_NO_RIDE; fi
    TGMX_STEPS=200 TGMX_ROUNDS=1 rocprofv3 --kernel-trace --stats -d $OUT/p_${WL}_${w}_$tag -o t -- python $ROOT/tools/time_step_world.py $WL ring $w > $OUT/p_${WL}_${w}_$tag.log 2>&1
    python $ROOT/tools/rocpd_summary.py $OUT/p_${WL}_${w}_$tag/t_results.db > $OUT/sum_${WL}_${w}_$tag.txt 2>&1
    rm -rf $OUT/p_${WL}_${w}_$tag
  done
done

ROOT=$PWD; OUT=$ROOT/gpurun_out/r02b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in 1 8; do
  for tag in ride noride; do
    if [ $tag = noride ]; then export TGMX_NO_RIDE=1; else unset TGMX_NO_RIDE; fi
    TGMX_STEPS=200 rocprofv3 --kernel-trace --stats -d $OUT/p_${w}_$tag -o t -- python $ROOT/tools/time_step_world.py wiki ring $w > $OUT/p_${w}_$tag.log 2>&1
    python $ROOT/tools/rocpd_summary.py $OUT/p_${w}_$tag/t_results.db > $OUT/sum_${w}_$tag.txt 2>&1
    rm -rf $OUT/p_${w}_$tag
  done
done

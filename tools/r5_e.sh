#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$PWD; O=$ROOT/gpurun_out/r5_e; mkdir -p $O
export TMPDIR=/tmp
for V in 1 ""; do
  T=$O/trace_s$V; rm -rf $T
  (cd /tmp && TGMX_BENCH_TGN_NO_LOADER_PASS=1 TGMX_BENCH_TGN_STREAMS=$V timeout -k 10 300 rocprofv3 --kernel-trace -d $T -- python $ROOT/tools/bench_tgn.py 300) > $O/trace_s$V.log 2>&1
  echo "streams='$V'"; grep '^{' $O/trace_s$V.log | tail -1 | cut -c200-420
  python tools/prof_summary.py union $T
  rm -rf $T
done

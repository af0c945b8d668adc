#!/bin/bash
# refresh of cfg 3's profiles after the last kernel change of the round (tgn_store_batch_kernel)
cd "$(dirname "$0")/.."
O=gpurun_out/profiles_r05b; rm -rf $O; mkdir -p $O
j() { grep '^{' | tail -1; }
for i in 1 2 3; do timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j >> $O/r05_bench_tgn.jsonl; done
for i in 1 2 3; do TGMX_BENCH_TGN_STREAMS=0 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j >> $O/r05_bench_tgn_one_stream.jsonl; done
TGMX_BENCH_TGN_PHASES=1 timeout 300 python tools/bench_tgn.py 400 2>/dev/null | j > $O/r05_bench_tgn_host_phases.json
TGMX_BENCH_TGN_STREAMS=0 TGMX_BENCH_TGN_NO_LOADER_PASS=1 tools/gpu_trace_byname.sh tgn 300 python $PWD/tools/bench_tgn.py 200 > $O/r05_tgn_rocprof_summary.md 2>/dev/null
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | j > $O/r05_bench_ring_driver_args_$i.json; done
python - <<'PY'
import json
O='gpurun_out/profiles_r05b'
L=lambda p:[json.loads(l) for l in open(p) if l.startswith('{')]
print('side', [(round(x['pipeline_us_per_batch'],1), round(x['host_busy_us_per_batch'],1)) for x in L(O+'/r05_bench_tgn.jsonl')])
print('one ', [(round(x['pipeline_us_per_batch'],1), round(x['host_busy_us_per_batch'],1)) for x in L(O+'/r05_bench_tgn_one_stream.jsonl')])
print(open(O+'/r05_bench_tgn_host_phases.json').read()[:500])
for i in (1,2,3):
    d=json.load(open(f'{O}/r05_bench_ring_driver_args_{i}.json')); print(round(d['ms_per_step']*1e3,2), round(d['value']/1e9,3), round(d['roofline']['frac'],3), d['roofline']['traffic'] and d['roofline']['traffic']['bytes'], {k:round(v['pipeline_us_per_batch'],1) for k,v in d['pipeline_cfg3'].items() if isinstance(v,dict)}, round(d['default_path']['ms_per_step']*1e3,1), round(d['default_path']['released_ms_per_step']*1e3,1))
PY

#!/bin/bash
# rocprofv3 --pmc <counters> of an arbitrary python command (counters only: never combined with a trace); prints per-kernel averages.
# Usage: tools/gpu_pmc_cmd.sh <tag> "<counter list>" <kernel substring> python /root/repo/...
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=$1; CTRS=$2; KERN=$3; shift 3
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 140 rocprofv3 --pmc $CTRS -d "$OUT" -- "$@") > "$OUT.log" 2>&1
python - "$OUT" "$KERN" <<'PY'
import glob, os, sqlite3, sys
out, kern = sys.argv[1], sys.argv[2]
dbs = sorted(glob.glob(os.path.join(out, '**', '*.db'), recursive=True))
if not dbs:
    print('no .db under', out); sys.exit(0)
db = sqlite3.connect(dbs[-1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
pm = [t for t in tabs if t.startswith('pmc_events') or t == 'pmc_events']
ke = [t for t in tabs if t.startswith('kernels') or t == 'kernels']
try:
    rows = db.execute("select k.name, k.grid_x, p.counter_name, count(*), avg(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
                      "where k.name like ? group by k.name, k.grid_x, p.counter_name order by k.name, k.grid_x", ('%' + kern + '%',)).fetchall()
    for r in rows:
        print(f'{r[0][:60]:60s} grid={r[1]:>8} {r[2]:36s} n={r[3]:>5} avg={r[4]:.1f}')
except Exception as e:
    print('query failed:', e, tabs[:40])
PY
rm -rf "$OUT"

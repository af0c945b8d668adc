#!/bin/bash
# rocprofv3 --memory-copy-trace of a python command: the copies by (direction, size) -- who is copying what?   tools/gpu_memcopies.sh <tag> python ...
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
TAG=$1; shift
OUT=$ROOT/gpurun_out/mc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --memory-copy-trace --kernel-trace -d "$OUT" -- "$@") > "$OUT.log" 2>&1
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
dbs = sorted(glob.glob(os.path.join(sys.argv[1], '**', '*.db'), recursive=True))
db = sqlite3.connect(dbs[-1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cand = [t for t in tabs if 'memory_cop' in t.lower() or 'memcpy' in t.lower()]
print('tables:', cand)
for t in cand[:3]:
    cols = [r[1] for r in db.execute(f'pragma table_info({t})')]
    print(t, cols)
    try:
        size_col = next(c for c in cols if 'size' in c.lower() or 'bytes' in c.lower())
        name_col = next((c for c in cols if c.lower() in ('name', 'kind', 'direction')), cols[0])
        for r in db.execute(f'select {name_col}, {size_col}, count(*), avg(end - start) from {t} group by {name_col}, {size_col} order by count(*) desc limit 25'):
            print(r)
    except Exception as e:
        print('query failed', e)
PY
rm -rf "$OUT"

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
for r in db.execute('select name, size, count(*), avg(duration), min(start) from memory_copies group by name, size order by count(*) desc limit 30'):
    print(f'{r[0]:28s} size={r[1]:>10} n={r[2]:>6} avg={r[3] / 1e3:8.2f} us')
PY
rm -rf "$OUT"

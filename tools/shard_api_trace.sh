#!/bin/bash
# HIP runtime API call counts of one stream of the view-sharded path at world 1 against the unsharded path
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-shardapi}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MODSX_PAIR_NOSPLIT=1 MODSX_PAIR_SERIAL=1
ONE="--steps 2 --warmup 1 --workers 1 --batch 4 --no-cpu-baseline --no-extra"
for mode in unsharded sharded; do
  a=""; [ $mode = sharded ] && a="--shard views"
  rm -rf /tmp/rq_$mode
  rocprofv3 --hip-runtime-trace --kernel-trace -d /tmp/rq_$mode -o p -- python $R/bench.py $ONE $a > /tmp/rq_$mode.log 2>&1
  DB=$(find /tmp/rq_$mode -name "*.db" | head -1)
  python - $DB > $OUT/api_$mode.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if not t.startswith("rocpd_") or t.count("_") < 3])
for t in ("regions", "regions_and_samples", "api"):
    if t in tabs:
        cols = [c[1] for c in db.execute("pragma table_info(%s)" % t)]
        print(t, cols)
        for r in db.execute("select name, count(*) from %s group by name order by count(*) desc limit 30" % t):
            print(r)
        break
PY
done

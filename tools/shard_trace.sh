#!/bin/bash
# kernel + memory-copy trace of one stream of the view-sharded path at world 1 (what the path adds to the unsharded kernels)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-shardtrace}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MODSX_PAIR_NOSPLIT=1 MODSX_PAIR_SERIAL=1
ONE="--steps 2 --warmup 1 --workers 1 --batch 4 --no-cpu-baseline --no-extra"
for mode in unsharded sharded; do
  a=""; [ $mode = sharded ] && a="--shard views"
  rm -rf /tmp/rp_$mode
  rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/rp_$mode -o p -- python $R/bench.py $ONE $a > /tmp/rp_$mode.log 2>&1
  DB=$(find /tmp/rp_$mode -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py $DB $OUT/kernels_$mode.txt "bench.py $ONE $a" > /dev/null
  python - $DB > $OUT/copies_$mode.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if "memory_cop" in x.lower()]
print("tables:", t)
for name in t[:3]:
    cols = [c[1] for c in db.execute("pragma table_info(%s)" % name)]
    print(name, cols)
    if name != "memory_copies":
        continue
    for r in db.execute("select name, count(*), sum(size), sum(duration), avg(duration) from memory_copies group by name"):
        print(r)
    print("by direction and size (bytes): count, total us")
    for r in db.execute("select name, size, count(*), sum(duration) / 1000 from memory_copies group by name, size having count(*) >= 8 order by count(*) desc limit 40"):
        print(r)
PY
done
tail -3 /tmp/rp_sharded.log

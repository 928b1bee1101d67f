#!/bin/bash
# fraction of wall time during which at least one kernel is running (default multi-context run)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_busy
rocprofv3 --kernel-trace -d /tmp/rp_busy -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /tmp/rp_busy.log 2>&1
DB=$(find /tmp/rp_busy -name "*.db" | head -1)
python - <<PY
import sqlite3
db = sqlite3.connect("$DB")
rows = db.execute("select start, end from kernels order by start").fetchall()
# last 60 % of the trace = timed region (roughly); compute union coverage and average concurrency
t0 = rows[0][0]; t1 = max(r[1] for r in rows)
lo = t0 + 0.35 * (t1 - t0); hi = t0 + 0.9 * (t1 - t0)
cov = 0; cur_s = None; cur_e = None; tot = 0
for s, e in rows:
    if e < lo or s > hi: continue
    s = max(s, lo); e = min(e, hi); tot += e - s
    if cur_e is None or s > cur_e:
        if cur_e is not None: cov += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None: cov += cur_e - cur_s
print("window %.1f ms  busy(union) %.3f  avg concurrency %.2f" % ((hi - lo) / 1e6, cov / (hi - lo), tot / (hi - lo)))
PY

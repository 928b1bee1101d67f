#!/usr/bin/env python3
"""Kernel sequence of a rocprofv3 rocpd database: start (us since the first kernel), duration, grid, name -- for reading the
launch structure of one launch set (gaps = host time, short kernels = launch-bound octaves)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
start = "start" if "start" in cols else "start_timestamp"
rows = db.execute("select %s, duration, grid_x, workgroup_x, name from kernels order by %s" % (start, start)).fetchall()
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
t0 = rows[lo][0]
prev_end = t0
for r in rows[lo:hi]:
    name = re.sub(r"^void ", "", r[4].split("(")[0]).replace("mx::", "")
    print("%9.1f  gap %6.1f  dur %7.1f  wgs %6d  %s" % ((r[0] - t0) / 1e3, (r[0] - prev_end) / 1e3, r[1] / 1e3, (r[2] or 0) // max(1, r[3] or 1), name[:40]))
    prev_end = r[0] + r[1]

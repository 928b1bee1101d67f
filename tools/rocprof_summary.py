#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into the text summary kept under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    raw = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    # template instantiations (k_blur_hess<R>, ...) are reported as one kernel
    import re
    merged = {}
    for r in raw:
        key = re.sub(r"<.*", "", re.sub(r"^void ", "", r[0].split("(")[0]))
        m = merged.get(key)
        if m is None:
            merged[key] = [key] + list(r[1:])
        else:
            m[1] += r[1]; m[2] += r[2]; m[4] = min(m[4], r[4]); m[5] = max(m[5], r[5])
            for q in (6, 7, 8, 9, 10, 11):
                m[q] = max(m[q] or 0, r[q] or 0)
            m[3] = m[2] / m[1]
    rows = sorted(merged.values(), key=lambda r: -r[2])
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
        if note:
            f.write("# " + note + "\n")
        f.write("%-28s %7s %12s %10s %10s %10s %6s %5s %5s %5s %7s\n" %
                ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds"))
        for r in rows:
            name = r[0].split("(")[0].replace("mx::", "")
            f.write("%-28s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %5d %7d\n" %
                    (name[:28], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
                     r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0))
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))

#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters (rocpd database) as a table."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                      "group by kernel_name, counter_name").fetchall()
    names = sorted({r[1] for r in rows})
    table = {}
    for k, c, n, v in rows:
        k = k.split("(")[0].replace("mx::", "")
        table.setdefault(k, {"calls": n})[c] = v
    with open(out_path, "w") as o:
        o.write("# rocprofv3 --pmc %s: per-dispatch averages\n" % " ".join(names))
        if note:
            o.write("# " + note + "\n")
        o.write("%-26s %6s " % ("kernel", "calls") + " ".join("%20s" % n[:20] for n in names) + "\n")
        for k in sorted(table, key=lambda k: -table[k].get(names[0], 0)):
            o.write("%-26s %6d " % (k[:26], table[k]["calls"]) + " ".join("%20.0f" % table[k].get(n, 0) for n in names) + "\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))

#!/usr/bin/env python3
"""DESIGN.md section 5's kernel table from the committed profiles of a round: usage design_tables.py r06 [pairs] -> markdown on stdout.
Columns: launches per pair, one-stream duration, ms per pair and share (profiles/<tag>_kernel_stats_single_stream.txt), duration under
the bench's 16 streams (<tag>_kernel_stats_default.txt), vector instructions per launch and VALU issue share
(SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 x 256), <tag>_pmc_sq.txt), HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, <tag>_pmc_hbm.txt)."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
pairs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0       # pairs of the one-stream profile run; default: the k_match_sweep1 launches of that run (one per pair)

def short(name):
    return re.sub(r"<.*", "", name.replace("void ", "").strip())

def stats(path):
    rows = {}
    if not os.path.exists(path):
        return rows
    for ln in open(path):
        if ln.startswith("#") or ln.startswith("kernel"):
            continue
        p = ln.split()
        if len(p) < 7:
            continue
        # name may hold spaces (templates): numeric fields are the last 10
        name = short(" ".join(p[:-10]))
        calls, total, avg = float(p[-10]), float(p[-9]), float(p[-8])
        r = rows.setdefault(name, [0.0, 0.0])
        r[0] += calls; r[1] += total
    return rows

def counters(path):
    rows, cols = {}, []
    if not os.path.exists(path):
        return rows, cols
    for ln in open(path):
        if ln.startswith("#"):
            continue
        if ln.startswith("kernel"):
            cols = ln.split()[2:]
            continue
        p = ln.rstrip().rsplit(None, len(cols) + 1)
        if len(p) != len(cols) + 2:
            continue
        name, calls = short(p[0]), float(p[1])
        vals = [float(x) for x in p[2:]]
        r = rows.setdefault(name, [0.0] + [0.0] * len(cols))
        r[0] += calls
        for i, v in enumerate(vals):
            r[1 + i] += v * calls
    return rows, cols

one = stats(os.path.join(ROOT, "profiles", tag + "_kernel_stats_single_stream.txt"))
dflt = stats(os.path.join(ROOT, "profiles", tag + "_kernel_stats_default.txt"))
sq, sqc = counters(os.path.join(ROOT, "profiles", tag + "_pmc_sq.txt"))
hbm, hbc = counters(os.path.join(ROOT, "profiles", tag + "_pmc_hbm.txt"))
tot = sum(v[1] for v in one.values())
if pairs <= 0:
    pairs = one.get("k_match_sweep1", [33.0, 0.0])[0]
print("| kernel | launches / pair | µs / launch, one stream | ms / pair, one stream | share | µs / launch under 16 streams | M vector instr. / launch | VALU issue | HBM MB / launch |")
print("|---|---|---|---|---|---|---|---|---|")
for name, (calls, total) in sorted(one.items(), key=lambda kv: -kv[1][1]):
    if total / tot < 0.002:
        continue
    d = dflt.get(name)
    row = [ "`%s`" % name, "%.1f" % (calls / pairs), "%.1f" % (total / calls), "%.3f" % (total / pairs / 1e3), "%.1f %%" % (100 * total / tot),
            "%.0f" % (d[1] / d[0]) if d and d[0] else "--" ]
    s = sq.get(name)
    if s and "SQ_INSTS_VALU" in sqc and "GRBM_GUI_ACTIVE" in sqc:
        iv, ga = s[1 + sqc.index("SQ_INSTS_VALU")] / s[0], s[1 + sqc.index("GRBM_GUI_ACTIVE")] / s[0]
        row += ["%.1f" % (iv / 1e6), "%.2f" % (iv / (ga / 8 * 256)) if ga else "--"]
    else:
        row += ["--", "--"]
    h = hbm.get(name)
    if h and len(hbc) >= 2:
        # pmc_summary.py columns: fetch, write, fetch x 2, write  (bytes per launch); the last two are the corrected ones
        row.append("%.1f" % ((h[-2] + h[-1]) / h[0] / 1e6))
    else:
        row.append("--")
    print("| " + " | ".join(row) + " |")
print("| **sum** | | | **%.2f** | | | | | |" % (tot / pairs / 1e3))

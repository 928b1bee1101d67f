#!/usr/bin/env python3
"""Workgroup timeline of k_match_sweep1 (needs a library built with -DMATCH_TRACE: tools/build_variant.sh trace "-DMATCH_TRACE",
MODSX_LIB=mods_amd/libmodsx_trace.so).  Prints when the workgroups of one launch start and end, and where they ran."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mods_amd
from mods_amd import synthetic

tilts = sys.argv[1] if len(sys.argv) > 1 else "1,2,4,6,8"
phi = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
ctx = mods_amd.Context(0)
a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
views = mods_amd.set_vs_pars([1.0], [float(t) for t in tilts.split(",")], phi, 0.2, 1, [])
params = mods_amd.default_pair_params()
ia, ib = ctx.upload(a), ctx.upload(b)
r1, d1 = ctx.detect_describe_views(ia, views, params)
r2, d2 = ctx.detect_describe_views(ib, views, params)
pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
for _ in range(3):
    ctx.match_fginn(d1, d2, pos2)
lib = mods_amd.lib()
n = 16384
buf = np.zeros((n, 4), np.uint64)
rc = lib.modsx_debug_match_trace(buf.ctypes.data_as(ctypes.c_void_p), n)
assert rc == 0, rc
live = buf[:, 0] > 0
t = buf[live]
t0 = t[:, 0].min()
st = (t[:, 0] - t0).astype(np.float64) / 100.0      # us (100 MHz)
en = (t[:, 1] - t0).astype(np.float64) / 100.0
cyc = t[:, 3].astype(np.float64)
hw = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
print("n1 %d n2 %d  workgroups %d" % (len(d1), len(d2), live.sum()))
print("start us: min %.1f p50 %.1f p90 %.1f max %.1f" % (st.min(), np.median(st), np.percentile(st, 90), st.max()))
print("end   us: min %.1f p50 %.1f p90 %.1f max %.1f" % (en.min(), np.median(en), np.percentile(en, 90), en.max()))
dur = en - st
print("life  us: min %.1f p50 %.1f p90 %.1f max %.1f   shader cycles p50 %.0f" % (dur.min(), np.median(dur), np.percentile(dur, 90), dur.max(), np.median(cyc)))
key = xcc * 1000 + se * 100 + sh * 20 + cu
u, c = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu): %d   workgroups per CU: %s" % (len(u), dict(zip(*np.unique(c, return_counts=True)))))
ux, cx = np.unique(xcc, return_counts=True)
print("per XCC:", dict(zip(ux.tolist(), cx.tolist())))
h, edges = np.histogram(st, bins=12)
print("start histogram:", h.tolist(), ["%.0f" % e for e in edges])
h, edges = np.histogram(en, bins=12)
print("end histogram:", h.tolist(), ["%.0f" % e for e in edges])
# concurrency over time
ts = np.linspace(0, en.max(), 16)
print("resident:", [(int(((st <= x) & (en > x)).sum())) for x in ts])

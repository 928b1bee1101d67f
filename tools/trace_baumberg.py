#!/usr/bin/env python3
"""Where a wavefront of k_baumberg_stream spends its time (a -DBAUM_TRACE build: tools/build_variant.sh btrace "-DBAUM_TRACE", then
MODSX_LIB=mods_amd/libmodsx_btrace.so python tools/trace_baumberg.py): shader-clock time per phase of the iteration loop, summed
over all wavefronts of the launches of a few 31-view image sets."""
import ctypes
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mods_amd
from mods_amd import synthetic

a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=5500, seed=12345)
ctx = mods_amd.Context(0)
views = mods_amd.set_vs_pars([1.0], [1.0, 2.0, 4.0, 6.0, 8.0], 120.0, 0.2, 1, [])
par = mods_amd.default_pair_params()
ia = ctx.upload(a)
ctx.detect_describe_views(ia, views, par)
lib = mods_amd.lib()
buf = (ctypes.c_ulonglong * 16)()
lib.modsx_debug_baum_trace(buf, 1)
for _ in range(3):
    ctx.detect_describe_views(ia, views, par)
lib.modsx_debug_baum_trace(buf, 0)
t = np.array(list(buf[:9]), np.float64)
names = ["refill", "coordinates (both slots)", "slot 0: job + border test + taps", "slot 0: gradients + products", "slot 1: job + border test + taps",
         "slot 1: gradients + products", "ordered sums (361 adds x 3, both slots)", "Jacobi + update + convergence"]
tot = t[:8].sum()
print("wavefronts %d, %.0f clocks per wavefront" % (t[8], tot / max(1, t[8])))
for n, v in zip(names, t[:8]):
    print("  %-42s %5.1f %%" % (n, 100 * v / tot))
h = np.array(list(buf[9:16]), np.float64)
print("window extent in the level (max of width, height; px) per slot-iteration: " +
      ", ".join("%s %.1f %%" % (k, 100 * v / max(1, h.sum())) for k, v in zip(("<=16", "<=24", "<=32", "<=40", "<=48", "<=64", ">64"), h)))

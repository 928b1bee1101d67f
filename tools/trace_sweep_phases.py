#!/usr/bin/env python3
"""Where a wavefront of k_match_sweep1 spends its cycles, stage by stage (needs the phase-trace build of the matcher:
tools/build_variant.sh ptrace "-DSWEEP_PHASE_TRACE", MODSX_LIB=mods_amd/libmodsx_ptrace.so; MODSX_SWEEP1_FAT=0 / 1 picks the shape)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mods_amd
from mods_amd import synthetic
ctx = mods_amd.Context(0)
a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
views = mods_amd.set_vs_pars([1.0], [1.0, 2.0, 4.0, 6.0, 8.0], 120.0, 0.2, 1, [])
params = mods_amd.default_pair_params()
ia, ib = ctx.upload(a), ctx.upload(b)
r1, d1 = ctx.detect_describe_views(ia, views, params)
r2, d2 = ctx.detect_describe_views(ib, views, params)
pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
for _ in range(3):
    ctx.match_fginn(d1, d2, pos2)
n = 16384
buf = np.zeros((n, 8), np.uint64)
assert mods_amd.lib().modsx_debug_phase_trace(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
t = buf[buf[:, 4] > 0].astype(np.float64)
st = t[:, 4]
print("n1 %d n2 %d: %d wavefronts traced, QSETS %d, stages per wavefront p50 %d" % (len(d1), len(d2), len(t), int(t[0, 6]), int(np.median(st))))
tot = t[:, 3]
life = (t[:, 5] - t[:, 7]) / 100.0          # us (the 100 MHz wall clock)
print("a wavefront lives %.1f us (p10 %.1f p90 %.1f), starts %.1f .. %.1f us, ends %.1f .. %.1f us after the first; shader clock over its life: %.2f GHz" % (
    np.median(life), np.percentile(life, 10), np.percentile(life, 90), (t[:, 7].min() - t[:, 7].min()) / 100, (t[:, 7].max() - t[:, 7].min()) / 100,
    (t[:, 5].min() - t[:, 7].min()) / 100, (t[:, 5].max() - t[:, 7].min()) / 100, np.median(tot / life) / 1e3))
print("cycles per wavefront: p50 %.0f  (per stage %.0f; a group of 4 tiles is %d MFMAs = %d matrix-pipe cycles)" % (np.median(tot), np.median(tot / st), 16 * int(t[0, 6]), 512 * int(t[0, 6])))
for name, col in (("waiting for the stage's DMA + barrier", 0), ("issuing the next stage's DMA", 1), ("the four tiles (LDS reads, MFMAs, epilogue)", 2)):
    print("  %-46s %5.1f %% of the wavefront's cycles, %7.0f cycles per stage (p10 %.0f p90 %.0f)" % (name, 100 * np.median(t[:, col] / tot), np.median(t[:, col] / st), np.percentile(t[:, col] / st, 10), np.percentile(t[:, col] / st, 90)))

#!/usr/bin/env python3
"""Host profile of the F verification (exp_ransacFcustom + DEGENSAC) on synthetic two-view problems: where a call's time goes.
Needs a library built with -DMODSX_HPROF (tools/build_variant.sh hprof "-DMODSX_HPROF"; MODSX_LIB=mods_amd/libmodsx_hprof.so).
Without a GPU the rFtH loop counts its hypotheses on the host (the device does that in the pipeline): the `rFtH` line minus
`innerFH` is that counting."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mods_amd
from common import synth_two_view

NAMES = {0: "innerFH: FDs(all)", 1: "innerFH: u2f(10 pts)", 2: "u2Fit (total)", 3: "u2Fit: FDs(all)", 4: "u2Fit: u2f(inliers)", 5: "innerFH: dual_sample",
         6: "innerFH (total)", 8: "u2f: normu + lin_fmN", 9: "u2f: cov_mat", 10: "u2f: jacobi 9x9", 11: "u2f: singulF + denorm", 12: "u2f: left_null9",
         13: "main loop: fds + inlidxs per hypothesis", 14: "innerH", 15: "rFtH (total)", 16: "lsq_and_lo", 17: "ransac_f (total)"}
n_in, n_out, pf = int(sys.argv[1]) if len(sys.argv) > 1 else 3200, int(sys.argv[2]) if len(sys.argv) > 2 else 3200, float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
L = mods_amd.lib()
out = (ctypes.c_long * 48)()
L.modsx_debug_hprof(out, 1)
t0 = time.time()
for seed in range(1, 7):
    pts, laf = synth_two_view(seed, n_in=n_in, n_out=n_out, planar_frac=pf)
    mods_amd.loransac_f(pts, laf, laf, seed=seed)
print("6 calls, %.0f ms" % (1e3 * (time.time() - t0)))
L.modsx_debug_hprof(out, 0)
for i in sorted(NAMES):
    if out[24 + i]:
        print("%-42s %9.1f ms %9d calls %9.1f us/call" % (NAMES[i], out[i] / 1e6, out[24 + i], out[i] / 1e3 / out[24 + i]))

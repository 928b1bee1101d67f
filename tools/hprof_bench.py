#!/usr/bin/env python3
"""bench.py under a -DMODSX_HPROF library (MODSX_LIB=mods_amd/libmodsx_hprof.so), then the host profile of the F verification
summed over all verifying threads.  usage: hprof_bench.py <bench.py arguments>"""
import ctypes, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mods_amd
NAMES = {0: "innerFH: FDs(all)", 1: "innerFH: u2f(10 pts)", 2: "u2Fit (total)", 3: "u2Fit: FDs(all)", 4: "u2Fit: u2f(inliers)", 5: "innerFH: dual_sample",
         6: "innerFH (total)", 8: "u2f: normu + lin_fmN", 9: "u2f: cov_mat", 10: "u2f: jacobi 9x9", 11: "u2f: singulF + denorm", 12: "u2f: left_null9",
         13: "main loop: fds + inlidxs per hypothesis", 14: "innerH", 15: "rFtH (total)", 16: "lsq_and_lo", 17: "ransac_f (total)"}
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
out = (ctypes.c_long * 48)()
mods_amd.lib().modsx_debug_hprof(out, 0)
for i in sorted(NAMES):
    if out[24 + i]:
        print("%-42s %10.1f ms %9d calls %9.1f us/call" % (NAMES[i], out[i] / 1e6, out[24 + i], out[i] / 1e3 / out[24 + i]))

#!/usr/bin/env python3
"""Stand-alone timing of the FGINN matcher (kernels_match.hip) at configs[2]-like sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mods_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rs = np.random.RandomState(0)
# SIFT-like descriptors: sparse-ish, norm ~512
def mk(n):
    d = rs.gamma(0.6, 30.0, (n, 128))
    d = d / np.linalg.norm(d, axis=1, keepdims=True) * 512
    return np.clip(np.floor(d), 0, 255).astype(np.float32)
d2 = mk(n)
d1 = d2[rs.permutation(n)] + rs.randint(-6, 7, (n, 128))          # noisy copies: realistic NN structure
d1 = np.clip(d1, 0, 255).astype(np.float32)
pos2 = rs.uniform(0, 1000, (n, 2))
ctx = mods_amd.Context(0)
t = ctx.match_fginn(d1, d2, pos2)            # warm-up (includes H2D of descriptors)
ctx.profile(True)
t0 = time.time()
for _ in range(reps):
    t = ctx.match_fginn(d1, d2, pos2)
dt = (time.time() - t0) / reps
st = ctx.kernel_stats()["match_fginn"]
ms = st["ms"] / reps
flops = 2.0 * n * n * 128
print("N=M=%d  tentatives %d  wall %.2f ms (incl. H2D of 2x%.1f MB)  matcher kernels %.3f ms  -> %.1f TFLOP/s algorithmic "
      "(2NM128 / time; %.2f%% of the 5 PFLOP/s dense int8 peak)" % (n, len(t), dt * 1e3, n * 128 / 1e6, ms, flops / ms / 1e9,
                                                                  100 * flops / ms / 1e9 / 5000))

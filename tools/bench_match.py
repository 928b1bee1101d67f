#!/usr/bin/env python3
"""Stand-alone timing of the FGINN matcher (kernels_match.hip) on REAL descriptors.

The descriptors are those of a synthetic 1024x768 pair under a view ladder (default: TiltSet 1,2,3,4,6, Phi 360 = 8 views,
~10 k regions per side; --tilts 1,2,4,6,8 --phi 120 gives 31 views, ~24 k per side), i.e. what the matcher sees in the
multi-view configurations: many near-duplicate trains per query, so many walks go past the second neighbour.
All matcher launches (pack, sweep 1, decide, resolve) are inside the timed bracket.
  --check N   compare the first N queries with the CPU oracle (test infrastructure, ~1.4 s per 1000 x 20 k)
  --rep K     descriptors replicated K times with +-1 noise on a few entries (bigger problems from the same statistics)
"""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mods_amd
from mods_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--tilts", default="1,2,3,4,6")
ap.add_argument("--phi", type=float, default=360.0)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rep", type=int, default=1)
ap.add_argument("--check", type=int, default=0)
ap.add_argument("--synthetic", type=int, default=0, help="N: random SIFT-like descriptors instead of real ones")
args = ap.parse_args()

ctx = mods_amd.Context(0)
if args.synthetic:
    rs = np.random.RandomState(0)
    def mk(n):
        d = rs.gamma(0.6, 30.0, (n, 128))
        d = d / np.linalg.norm(d, axis=1, keepdims=True) * 512
        return np.clip(np.floor(d), 0, 255).astype(np.float32)
    d2 = mk(args.synthetic)
    d1 = np.clip(d2[rs.permutation(len(d2))] + rs.randint(-6, 7, d2.shape), 0, 255).astype(np.float32)
    pos2 = rs.uniform(0, 1000, (len(d2), 2))
    what = "synthetic noisy copies"
else:
    a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
    views = mods_amd.set_vs_pars([1.0], [float(t) for t in args.tilts.split(",")], args.phi, 0.2, 1, [])
    params = mods_amd.default_pair_params()
    ia, ib = ctx.upload(a), ctx.upload(b)
    r1, d1 = ctx.detect_describe_views(ia, views, params)
    r2, d2 = ctx.detect_describe_views(ib, views, params)
    pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
    what = "%d views (tilts %s, phi %g)" % (len(views), args.tilts, args.phi)
    if args.rep > 1:
        rs = np.random.RandomState(1)
        def grow(d):
            out = [d]
            for k in range(1, args.rep):
                e = d.copy()
                idx = rs.randint(0, 128, (len(d), 4))
                e[np.arange(len(d))[:, None], idx] = np.clip(e[np.arange(len(d))[:, None], idx] + rs.randint(-1, 2, idx.shape), 0, 255)
                out.append(e)
            return np.concatenate(out)
        d1, d2 = grow(d1), grow(d2)
        pos2 = np.concatenate([pos2 + 0.25 * k for k in range(args.rep)])
n1, n2 = len(d1), len(d2)
t = ctx.match_fginn(d1, d2, pos2)            # warm-up (includes H2D of descriptors)
ctx.profile(True)
t0 = time.time()
for _ in range(args.reps):
    t = ctx.match_fginn(d1, d2, pos2)
dt = (time.time() - t0) / args.reps
st = ctx.kernel_stats()["match_fginn"]
ms = st["ms"] / args.reps
flops = 2.0 * n1 * n2 * 128
print("%s: N=%d M=%d  tentatives %d  wall %.2f ms  matcher kernels (all launches) %.3f ms  -> %.1f TFLOP/s algorithmic "
      "(2NM128 / time; %.2f%% of the 5 PFLOP/s dense int8 peak); bytes (N+M)*128 = %.1f MB -> %.3f TB/s"
      % (what, n1, n2, len(t), dt * 1e3, ms, flops / ms / 1e9, 100 * flops / ms / 1e9 / 5000, (n1 + n2) * 128 / 1e6,
         (n1 + n2) * 128 / ms / 1e9))
if args.check:
    from oracle import pyoracle as O
    k = min(args.check, n1)
    ref = O.match_fginn(d1[:k], d2, pos2)
    got = ctx.match_fginn(d1[:k], d2, pos2)
    ok = len(ref) == len(got) and all(np.array_equal(ref[f], got[f]) for f in ref.dtype.names)
    print("oracle check on the first %d queries: %s (%d tentatives)" % (k, "IDENTICAL" if ok else "MISMATCH", len(ref)))
    if not ok:
        sys.exit(1)

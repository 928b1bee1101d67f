#!/usr/bin/env python3
"""CPU sweep of the verification stage against the reference's compiled degensac (oracle/_ref): random H and F problems, small
(8-100 tentatives) and mid-size (200-3000), all error types.  Counts the problems whose trajectory (samples, LO count) or inlier /
kept sets differ.  usage: sweep_ransac_ref.py <problems per kind> <seed>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mods_amd
from common import synth_corr, synth_two_view
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as O

n, seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 1
assert O.ref_available(), "oracle/_ref is not built"
rng = np.random.default_rng(seed0)
bad = {"H small": 0, "H mid": 0, "F small": 0, "F mid": 0}
cnt = dict.fromkeys(bad, 0)
worstF = worstH = 0.0
t0 = time.time()
for case in range(n):
    for kind in ("small", "mid"):
        T = int(rng.integers(8, 100)) if kind == "small" else int(rng.integers(200, 3000))
        pts, laf, _ = synth_corr(T, float(rng.choice([0.15, 0.3, 0.5, 0.8])), noise=float(rng.choice([0.3, 0.7, 2.0])), seed=seed0 * 100000 + case)
        seed, et = int(rng.integers(1, 1000)), int(rng.integers(0, 3))
        a, b = O.loransac_h(pts, laf, laf, seed=seed, error_type=et), mods_amd.loransac_h(pts, laf, laf, seed=seed, error_type=et)
        same = (a["n"], a["samples"], a["lo_count"]) == (b["n"], b["samples"], b["lo_count"]) and np.array_equal(a["inl"], b["inl"]) and np.array_equal(a["keep"], b["keep"])
        cnt["H " + kind] += 1; bad["H " + kind] += not same
        if not same:
            print("  H %s case %d differs: T %d seed %d et %d: ref (n %d samples %d lo %d) here (n %d samples %d lo %d)" % (
                kind, case, T, seed, et, a["n"], a["samples"], a["lo_count"], b["n"], b["samples"], b["lo_count"]), flush=True)
        if same and a["n"] and abs(a["H"].ravel()[8]) > 1e-12 and abs(b["H"].ravel()[8]) > 1e-12:
            worstH = max(worstH, float(np.abs(a["H"] / a["H"].ravel()[8] - b["H"] / b["H"].ravel()[8]).max()))
        if kind == "small":
            n_in, n_out = int(rng.integers(8, 60)), int(rng.integers(0, 40))
        else:
            n_in, n_out = int(rng.integers(100, 1200)), int(rng.integers(50, 1200))
        pts, laf = synth_two_view(seed0 * 100000 + case, n_in=n_in, n_out=n_out, planar_frac=float(rng.choice([0, 0, 0.5, 0.9, 1.0])), noise=float(rng.choice([0.3, 1.0, 2.0])))
        seed, et = int(rng.integers(1, 1000)), int(rng.integers(0, 2))
        a = O.loransac_f(pts, laf, laf, err_threshold=4.0, laf_coef=3.0, seed=seed, error_type=et)
        b = mods_amd.loransac_f(pts, laf, laf, err_threshold=4.0, laf_coef=3.0, seed=seed, error_type=et)
        same = (a["n"], a["samples"], a["lo_count"]) == (b["n"], b["samples"], b["lo_count"]) and np.array_equal(a["inl"], b["inl"]) and np.array_equal(a["keep"], b["keep"])
        cnt["F " + kind] += 1; bad["F " + kind] += not same
        if not same:
            print("  F %s case %d differs: n_in %d n_out %d seed %d et %d: ref (n %d samples %d lo %d) here (n %d samples %d lo %d)" % (
                kind, case, n_in, n_out, seed, et, a["n"], a["samples"], a["lo_count"], b["n"], b["samples"], b["lo_count"]), flush=True)
        if same and a["n"]:
            Fa, Fb = a["F"] / np.linalg.norm(a["F"]), b["F"] / np.linalg.norm(b["F"])
            if (Fa * Fb).sum() < 0: Fb = -Fb
            worstF = max(worstF, float(np.abs(Fa - Fb).max()))
print("sweep_ransac_ref: %s ; differing: %s ; max |dH| %.3g  max |dF| %.3g ; %.0f s" % (cnt, bad, worstH, worstF, time.time() - t0))

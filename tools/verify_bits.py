#!/usr/bin/env python3
"""Digest of the verification stage's outputs (H / F matrices bit for bit, inlier and kept sets, sample counts) over a fixed set of
random problems: two builds of the library that print the same digest compute the same bits.  usage: verify_bits.py [problems]
(MODSX_LIB selects the library)"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mods_amd
from common import synth_corr, synth_two_view
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(77)
h = hashlib.sha256()
for case in range(n):
    T = int(rng.integers(8, 100)) if case % 2 else int(rng.integers(200, 4000))
    pts, laf, _ = synth_corr(T, float(rng.choice([0.15, 0.3, 0.5, 0.8])), noise=float(rng.choice([0.3, 0.7, 2.0])), seed=case)
    r = mods_amd.loransac_h(pts, laf, laf, seed=int(rng.integers(1, 1000)), error_type=int(rng.integers(0, 3)))
    h.update(np.ascontiguousarray(r["H"]).tobytes()); h.update(r["inl"].tobytes()); h.update(r["keep"].tobytes()); h.update(str((r["n"], r["samples"], r["lo_count"])).encode())
    n_in, n_out = (int(rng.integers(8, 60)), int(rng.integers(0, 40))) if case % 2 else (int(rng.integers(100, 3000)), int(rng.integers(50, 2000)))
    pts, laf = synth_two_view(case, n_in=n_in, n_out=n_out, planar_frac=float(rng.choice([0, 0.5, 0.9, 1.0])), noise=float(rng.choice([0.3, 1.0, 2.0])))
    r = mods_amd.loransac_f(pts, laf, laf, err_threshold=4.0, laf_coef=3.0, seed=int(rng.integers(1, 1000)), error_type=int(rng.integers(0, 2)))
    h.update(np.ascontiguousarray(r["F"]).tobytes()); h.update(r["inl"].tobytes()); h.update(r["keep"].tobytes()); h.update(str((r["n"], r["samples"], r["lo_count"])).encode())
print("verify_bits:", n, "problems of each kind:", h.hexdigest()[:32])

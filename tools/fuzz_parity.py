#!/usr/bin/env python3
"""Randomised end-to-end parity sweep (GPU box): many synthetic pairs of random size / content / blob density go through
modsx_match_pair and through the oracle; region counts, descriptors' consequences (tentatives, field by field), duplicate
filtering and -- where oracle/_ref is built -- RANSAC inliers and H must agree.  Not part of the pytest suites (the
oracle needs ~0.1-1 s per pair); run as `python tools/fuzz_parity.py [n_pairs] [seed0]`; `python tools/fuzz_parity.py views [n_images] [seed0]` sweeps the
multi-view loop instead (random tilt sets / rotation steps / blur: regions and descriptors of every view, bit for bit)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mods_amd  # noqa: E402
from mods_amd import synthetic  # noqa: E402
from common import oracle_pair, normH, same_records  # noqa: E402
from oracle import pyoracle  # noqa: E402


def fuzz_views(n, seed0):
    mods_amd.build()
    pyoracle.lib()
    O = pyoracle
    ctx = mods_amd.Context(0)
    rs = np.random.RandomState(seed0)
    bad = 0
    t0 = time.time()
    tilt_sets = ([1, 2], [1, 3, 5], [1, 2, 4, 8], [2, 6], [1, 2, 3, 4, 6], [1.5, 2.5])
    for i in range(n):
        rows, cols = int(rs.randint(70, 320)), int(rs.randint(70, 400))
        a, _, _ = synthetic.make_pair(rows=rows, cols=cols, nblobs=int(rs.randint(30, 500)), seed=int(rs.randint(1, 1 << 30)))
        tilts = [float(t) for t in tilt_sets[int(rs.randint(len(tilt_sets)))]]
        phi = float(rs.choice([360.0, 180.0, 72.0, 50.0]))
        sigma = float(rs.choice([0.2, 0.5, 0.8]))
        vo = O.set_vs_pars([1.0], tilts, phi, sigma, 1, [])
        vm = mods_amd.set_vs_pars([1.0], tilts, phi, sigma, 1, [])
        im = ctx.upload(a)
        rr, dr = O.detect_describe_views(a, vo)
        rg, dg = ctx.detect_describe_views(im, vm, mods_amd.default_pair_params())
        ok = len(rr) == len(rg) and np.array_equal(dr, dg) and same_records(rg, rr.view(mods_amd.REGION))
        if not ok:
            bad += 1
            print("MISMATCH image %d: %dx%d tilts %s phi %g sigma %g: regions %d vs %d" % (i, rows, cols, tilts, phi, sigma,
                                                                                         len(rg), len(rr)), flush=True)
        im.free()
    print("fuzz views: %d images, %d mismatches, %.1f s" % (n, bad, time.time() - t0))
    return 1 if bad else 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "views":
        return fuzz_views(int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(sys.argv[3]) if len(sys.argv) > 3 else 500)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    mods_amd.build()
    pyoracle.lib()
    O = pyoracle
    ctx = mods_amd.Context(0)
    rs = np.random.RandomState(seed0)
    bad = 0
    t0 = time.time()
    for i in range(n):
        rows, cols = int(rs.randint(60, 420)), int(rs.randint(60, 520))
        nbl = int(rs.randint(20, 700))
        seed = int(rs.randint(1, 1 << 30))
        a, b, _ = synthetic.make_pair(rows=rows, cols=cols, nblobs=nbl, seed=seed)
        if i % 5 == 4:   # flat-ish / noisy variants
            a = (a * 0.25 + 90).astype(np.float32)
        if i % 7 == 6:
            b = np.clip(b + rs.normal(0, 6, b.shape), 0, 255).astype(np.float32)
        ia, ib = ctx.upload(a), ctx.upload(b)
        rseed = int(rs.randint(1, 1000))
        got = ctx.match_pair(ia, ib, mods_amd.default_pair_params(ransac_seed=rseed))
        ref = oracle_pair(O, a, b, seed=rseed)
        ok = got["n_regions"] == (len(ref["d1"]), len(ref["d2"])) and got["n_tentatives"] == len(ref["tent"]) \
            and got["n_unique"] == len(ref["uniq"])
        if ok:
            for f in ref["uniq"].dtype.names:
                ok = ok and np.array_equal(got["tentatives"][f], ref["uniq"][f])
        if ok and ref.get("ransac") is not None:
            rr = ref["ransac"]
            okr = np.array_equal(got["ransac_inlier"], rr["inl"]) and np.array_equal(got["verified"], rr["keep"])
            if okr and rr["n"] > 0:
                okr = np.abs(normH(got["H"]) - normH(rr["H"])).max() < 1e-4
            # (until round 4 results that started a local optimisation from 8-9 inliers were excluded here: the reference's
            # 4-point u2h branch is now restated as it computes, see ransac_common.hpp u2h -- nothing is excluded any more)
            ok = okr
        if not ok:
            bad += 1
            print("MISMATCH pair %d: %dx%d blobs %d seed %d: regions %s vs %s, tentatives %d vs %d" %
                  (i, rows, cols, nbl, seed, got["n_regions"], (len(ref["d1"]), len(ref["d2"])), got["n_tentatives"],
                   len(ref["tent"])), flush=True)
        ia.free(); ib.free()
    print("fuzz: %d pairs, %d mismatches (regions, tentatives, duplicate filter, RANSAC inliers, verified set, H; no exclusions), %.1f s" %
          (n, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

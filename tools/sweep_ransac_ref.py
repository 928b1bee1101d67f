#!/usr/bin/env python3
"""CPU sweep of the verification stage against the reference's compiled degensac (oracle/_ref): random H and F problems, small
(8-100 tentatives) and mid-size (200-3000), all error types.  Counts the problems whose trajectory (samples, LO count) or inlier /
kept sets differ.  usage: sweep_ransac_ref.py <problems per kind> <seed> [processes]
The generator and the runner are tests/ransac_sweep.py (tests/test_ransac_sweep_cpu.py runs two sweeps of it in the CPU suite)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ransac_sweep as S
import pyoracle as O

n, seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 1
procs = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, min(8, len(os.sched_getaffinity(0))))
assert O.ref_available(), "oracle/_ref is not built"
t0 = time.time()
ps = S.problems(n, seed0)
rs = S.run_parallel(ps, procs)
cnt, bad, worst = {}, {}, {"H": 0.0, "F": 0.0}
for p, r in zip(ps, rs):
    k = "%s %s" % (p["kind"], p["size"])
    cnt[k] = cnt.get(k, 0) + 1
    bad[k] = bad.get(k, 0) + (not r["same"])
    if not r["same"]:
        print("  %s differs: %s: ref (n %d samples %d) here (n %d samples %d), %d inlier flags differ" % (
            r["name"], {k2: v for k2, v in p.items() if k2 not in ("kind", "size", "case", "seed0")}, r["n_ref"], r["samples_ref"],
            r["n_here"], r["samples_here"], r["inl_diff"]), flush=True)
    else:
        worst[p["kind"]] = max(worst[p["kind"]], r["d"])
print("sweep_ransac_ref: %s ; differing: %s ; max |dH| %.3g  max |dF| %.3g ; %.0f s on %d processes" % (
    cnt, bad, worst["H"], worst["F"], time.time() - t0, procs))

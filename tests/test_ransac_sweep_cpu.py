"""The verification stage against the reference's compiled degensac (oracle/_ref) on a SWEEP of random problems -- what
tools/sweep_ransac_ref.py ran by hand until round 5 is part of the CPU suite now: >= 1000 H and >= 1000 F problems (8-3000
tentatives, all error types, a quarter of the F scenes planar, i.e. DEGENSAC's plane-and-parallax branch and the near-degenerate
9 x 9 eigenproblems of its refits), the reference and libmodsx's host C++ run with the same seed, compared on trajectory (sample and
LO counts), inlier set and kept set.

The allowance is explicit: KNOWN lists the problems that may differ and how.  Anything else that differs fails the suite -- in
particular the counter-examples of the Jacobi solver's unsafe stopping rule (ransac_common.hpp, RULE 1: building libmodsx with
-DJACOBI_RULE=1 makes RULE1_COUNTER_EXAMPLES below differ; they were found by running this sweep against such a build).
"""
import os

import pytest

import ransac_sweep as S
from common import need_ref

# Two explicit allowances, nothing else:
#  1. KNOWN -- (kind, size, seed0, case) of problems on which the reference (stable from process to process) and libmodsx end a
#     near-tie apart, with the reason;
#  2. problems on which the REFERENCE DISAGREES WITH ITSELF between processes (ransac_sweep.reference_signatures: a differing
#     problem is re-run through the reference in six fresh processes; more than one distinct result, or one that equals libmodsx's,
#     means the comparison has no fixed right-hand side).  They are counted and bounded: at most 3 per 1000 problems.
KNOWN = {
    ("F", "mid", 4, 998): "134 inliers among 644 tentatives of a planar scene, the sampler runs to its cap of 10^5 samples: the reference's "
                          "LAPACK (dsyev_ of scipy's OpenBLAS here) and the Jacobi solver agree on an ill-conditioned null vector to ~1e-8 "
                          "and one tentative sits that close to the threshold -- 144 inliers against 145 (DESIGN.md section 8)",
}
# problems on which a libmodsx built with -DJACOBI_RULE=1 leaves the reference's trajectory while the shipped rule keeps it and the
# reference is stable (sweep seed0, case, kind, size): 549 inliers against 550, 30 flags apart -- the "another F, 30 other inliers"
# of ransac_common.hpp
RULE1_COUNTER_EXAMPLES = [(2, 1189, "F", "mid")]
REFERENCE_UNSTABLE_SEEN = [(4, 1051, "F", "small"), (2, 811, "F", "small")]   # always in the planar test's set: they exercise allowance 2


def _procs():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(8, n))


def _check(ps, rs):
    bad, unstable = [], []
    for p, r in zip(ps, rs):
        key = (p["kind"], p["size"], p["seed0"], p["case"])
        if r["same"]:
            continue
        if key in KNOWN:
            assert r["samples_ref"] == r["samples_here"] and abs(r["n_ref"] - r["n_here"]) <= 1, (r, KNOWN[key])
            continue
        sigs = S.reference_signatures(p, tries=6) | {r["sig_ref"]}
        if len(sigs) > 1 or r["sig_here"] in sigs:
            unstable.append(r["name"])
            continue
        bad.append(r)
    assert not bad, "problems that left the (stable) reference's trajectory / inlier set: %s" % bad
    assert len(unstable) <= max(2, 3 * len(ps) // 1000), "too many problems on which the reference is not reproducible: %s" % unstable
    return unstable


def test_sweep_h_and_f_against_the_reference_degensac(oracle):
    need_ref(oracle)
    # sweep (275, 7): 550 H + 550 F problems; sweep (250, 11): 500 + 500 -- 1050 H and 1050 F in all
    ps = S.problems(275, 7) + S.problems(250, 11)
    assert sum(p["kind"] == "H" for p in ps) >= 1000 and sum(p["kind"] == "F" for p in ps) >= 1000
    _check(ps, S.run_parallel(ps, _procs()))


def test_sweep_planar_family_and_the_known_cases(oracle):
    """The planar / close-eigenvalue family (where a stopping rule of the eigen-solver matters), the one listed allowance and
    the problems that caught the unsafe rule."""
    need_ref(oracle)
    ps = [p for p in S.problems(420, 4) if p["kind"] == "F" and p["planar"] >= 0.9]
    extra = {(k[2], k[3], k[0], k[1]) for k in KNOWN} | set(RULE1_COUNTER_EXAMPLES) | set(REFERENCE_UNSTABLE_SEEN)
    for seed0 in sorted({e[0] for e in extra}):
        top = max(e[1] for e in extra if e[0] == seed0) + 1
        ps += [p for p in S.problems(top, seed0) if (p["seed0"], p["case"], p["kind"], p["size"]) in extra and p not in ps]
    assert len(ps) >= 300
    rs = S.run_parallel(ps, _procs())
    _check(ps, rs)
    ran = {(p["seed0"], p["case"], p["kind"], p["size"]) for p in ps}
    assert extra <= ran

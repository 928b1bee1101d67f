"""DEGENSAC's plane-and-parallax hypothesis loop (rFtH, DegUtils.c:254-440) with its counting on the device (ransac_f.cpp:
k_rfth_count): the verifier's trajectory must stay the reference's.  The same cases run through the host loop in
tests/test_host_abi.py (no device in that container); here a device is present, so modsx.loransac_f takes the batched path."""
import numpy as np
import pytest

from common import synth_two_view, synth_corr, need_ref

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert (a["n"], a["samples"], a["lo_count"]) == (b["n"], b["samples"], b["lo_count"])
    assert np.array_equal(a["inl"], b["inl"]) and np.array_equal(a["keep"], b["keep"])
    Fa, Fb = a["F"] / max(np.linalg.norm(a["F"]), 1e-300), b["F"] / max(np.linalg.norm(b["F"]), 1e-300)
    if (Fa * Fb).sum() < 0:
        Fb = -Fb
    assert np.abs(Fa - Fb).max() < 1e-9


@pytest.mark.parametrize("planar_frac", [0.0, 0.5, 0.8, 1.0])
@pytest.mark.parametrize("error_type", [0, 1])
def test_degensac_with_device_counted_hypotheses_follows_the_reference(modsx, oracle, ctx, planar_frac, error_type):
    """Scenes from general to a single plane (every sample H-degenerate: rFtH on each): sample count, LO count, degeneracy count,
    inlier set, LAF-checked set and F of the reference's compiled degensac; on the planar scenes the device did the counting."""
    need_ref(oracle)
    modsx.verify_device_stats(reset=True)
    for seed in (1, 2, 3, 4, 5, 6):
        pts, laf = synth_two_view(seed, planar_frac=planar_frac, n_in=300 + 40 * seed, n_out=100 + 25 * seed)
        a = oracle.loransac_f(pts, laf, laf, seed=seed, error_type=error_type)
        b = modsx.loransac_f(pts, laf, laf, seed=seed, error_type=error_type)
        _same(a, b)
    st = modsx.verify_device_stats()
    assert st["disagreements"] == 0
    if planar_frac >= 0.5:
        assert st["batches"] > 0 and st["hypotheses"] >= st["batches"], st


def test_homography_scene_is_the_loops_worst_case(modsx, oracle, ctx):
    """A pure homography with outliers (what bench.py --config wxbs verifies with useF): no two-point epipole ever beats the
    plane, so every rFtH call runs its full 2 x 10^4 hypotheses -- all of them counted on the device, none changing the state."""
    need_ref(oracle)
    modsx.verify_device_stats(reset=True)
    for seed in (3, 8):
        pts, laf, _ = synth_corr(600, 0.7, noise=0.5, seed=seed)
        a = oracle.loransac_f(pts, laf, laf, seed=seed)
        b = modsx.loransac_f(pts, laf, laf, seed=seed)
        _same(a, b)
    st = modsx.verify_device_stats()
    assert st["disagreements"] == 0 and st["hypotheses"] > 10000, st


def test_many_off_plane_correspondences_go_through_several_lds_tiles(modsx, oracle, ctx):
    """More than 512 correspondences off the dominant plane: k_rfth_count stages them tile by tile; a plane with a few dozen
    inliers among 1500 random pairs also makes state-changing hypotheses frequent (each re-draws the rest of the loop)."""
    need_ref(oracle)
    modsx.verify_device_stats(reset=True)
    for seed, frac in ((11, 0.6), (12, 0.9)):
        pts, laf = synth_two_view(seed, planar_frac=frac, n_in=400, n_out=1500)
        a = oracle.loransac_f(pts, laf, laf, seed=seed, max_samples=20000)
        b = modsx.loransac_f(pts, laf, laf, seed=seed, max_samples=20000)
        _same(a, b)
    st = modsx.verify_device_stats()
    assert st["disagreements"] == 0 and st["batches"] > 0, st


def test_every_device_count_of_a_batch_equals_the_host_count(modsx, oracle, ctx, monkeypatch):
    """MODSX_VERIFY_DEVICE_CHECK=1: every hypothesis of every batch is counted again by the host's FDs -- not only the ones the
    device flags.  A device under-count (different f64 rounding in sqrt / division / operation order) would otherwise skip a state
    change without a trace; here it would show up as a disagreement."""
    need_ref(oracle)
    monkeypatch.setenv("MODSX_VERIFY_DEVICE_CHECK", "1")
    modsx.verify_device_stats(reset=True)
    for seed, frac, n_out in ((1, 0.8, 150), (2, 1.0, 200), (11, 0.6, 1500)):
        pts, laf = synth_two_view(seed, planar_frac=frac, n_in=350, n_out=n_out)
        a = oracle.loransac_f(pts, laf, laf, seed=seed, max_samples=20000)
        b = modsx.loransac_f(pts, laf, laf, seed=seed, max_samples=20000)
        _same(a, b)
    st = modsx.verify_device_stats()
    assert st["disagreements"] == 0 and st["hypotheses"] > 20000, st


def test_sweep_f_problems_with_the_device_counted_loop(modsx, oracle, ctx):
    """The -m gpu twin of tests/test_ransac_sweep_cpu.py: the F problems of two sweeps (8-2400 correspondences, a quarter planar) with a
    device present, i.e. rFtH's hypothesis counting on k_rfth_count wherever the loop goes quiet -- same allowances (the one KNOWN
    problem, problems on which the reference disagrees with itself between processes), plus the counter-example of the unsafe
    eigen-solver rule."""
    need_ref(oracle)
    import ransac_sweep as S
    from test_ransac_sweep_cpu import _check, RULE1_COUNTER_EXAMPLES
    modsx.verify_device_stats(reset=True)
    ps = [p for p in S.problems(60, 7) if p["kind"] == "F"] + [p for p in S.problems(160, 4) if p["kind"] == "F" and p["planar"] >= 0.5]
    for seed0, case, kind, size in RULE1_COUNTER_EXAMPLES:
        ps += [p for p in S.problems(case + 1, seed0) if (p["case"], p["kind"], p["size"]) == (case, kind, size)]
    assert len(ps) >= 250
    rs = [S.run_problem(p, mods=modsx, oracle=oracle) for p in ps]
    _check(ps, rs)
    st = modsx.verify_device_stats()
    assert st["disagreements"] == 0 and st["batches"] > 0 and st["hypotheses"] > 10000, st

"""GPU: parity of the HIP path (through the C ABI of libmodsx.so) with the CPU oracle, stage by stage.

Bar: bit-exact.  Every stage of this path ends in discrete decisions (extrema, thresholds, quantised
descriptors, ratio tests, RANSAC samples), so the f32/f64 intermediates are compared for exact equality too.
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from common import oracle_features, oracle_pair, laf_of, normH, same_records, need_ref

pytestmark = pytest.mark.gpu


def _rand_img(rows, cols, seed):
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:rows, 0:cols]
    img = 128 + 60 * np.sin(x / 7.0) * np.cos(y / 5.0) + rs.uniform(-30, 30, (rows, cols))
    return np.clip(np.floor(img), 0, 255).astype(np.float32)


def test_native_library_is_loaded(modsx, ctx):
    import os
    maps = open("/proc/self/maps").read()
    assert "libmodsx.so" in maps
    assert os.path.exists(modsx.LIB_PATH)


def test_gray_conversion(ctx, oracle, cat_pair):
    cat, cat2, _ = cat_pair
    for bgr in (cat, cat2):
        im = ctx.upload(bgr)
        assert np.array_equal(im.download(), oracle.gray_from_bgr(bgr))
        im.free()
    g = _rand_img(50, 70, 1)
    im = ctx.upload(g)
    assert np.array_equal(im.download(), g)
    im.free()


@pytest.mark.parametrize("shape", [(97, 131), (64, 64), (33, 200), (13, 17)])
@pytest.mark.parametrize("sigma", [0.7, 1.2263, 1.5199, 1.9466, 2.4525])
def test_gaussian_blur_bit_exact(ctx, oracle, shape, sigma):
    img = _rand_img(shape[0], shape[1], 3)
    im = ctx.upload(img)
    got = ctx.gaussian_blur(im, sigma)
    im.free()
    assert np.array_equal(got, oracle.gaussian_blur(img, sigma))


@pytest.mark.parametrize("det", [0, 1, 2])
@pytest.mark.parametrize("shape,norm", [((97, 131), 2.56), ((64, 64), 4.0637), ((33, 200), 10.24), ((240, 320), 1.0), ((5, 7), 2.56)])
def test_detector_responses_bit_exact(ctx, oracle, det, shape, norm):
    """ScaleSpaceDetector::Response for Hessian / DoG / Harris (pyramid.cpp:132-305) vs the oracle, every pixel (the Hessian
    frame is written 0 on both sides).  norm = sigma^2 of a pyramid level as at the call sites (pyramid.cpp:475, 490)."""
    rs = np.random.RandomState(det * 100 + shape[0])
    img = np.floor(rs.uniform(0, 255, shape)).astype(np.float32)
    im = ctx.upload(img)
    got = ctx.response(im, det, norm)
    im.free()
    ref = oracle.response(img, det, norm)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("shape", [(96, 128), (97, 131), (75, 125), (38, 63), (13, 14)])
def test_resize_half_bit_exact(ctx, oracle, shape):
    img = _rand_img(shape[0], shape[1], 4)
    im = ctx.upload(img)
    got = ctx.resize_half(im)
    im.free()
    ref = oracle.resize_half(img)
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.parametrize("shape", [(120, 160), (101, 77)])
def test_octave_levels_bit_exact(ctx, modsx, oracle, shape):
    img = _rand_img(shape[0], shape[1], 5)
    im = ctx.upload(img)
    blurs, resps = ctx.octave_levels(im, modsx.default_hessaff_params())
    im.free()
    rb, rr = oracle.octave_levels(img, oracle.default_params())
    assert np.array_equal(blurs, rb)
    assert np.array_equal(resps[:, 1:-1, 1:-1], rr[:, 1:-1, 1:-1])


def _check_sskp(a, b):
    assert len(a) == len(b)
    for f in ("octave", "level", "r0", "c0", "r", "c", "type", "b0", "b1", "b2", "val", "x", "y", "s", "pixelDistance"):
        assert np.array_equal(a[f], b[f]), f


@pytest.mark.parametrize("mode", [0, 4])
def test_scalespace_keypoints_bit_exact(ctx, modsx, oracle, small_pair, mode):
    for img in small_pair[:2]:
        im = ctx.upload(img)
        got = ctx.detect_scalespace(im, modsx.default_hessaff_params(mode=mode))
        im.free()
        ref = oracle.detect_scalespace(img, oracle.default_params(mode=mode))
        assert len(ref) > 50
        _check_sskp(got, ref)


@pytest.mark.parametrize("det,th", [(1, 1.5), (2, 400.0)])
def test_dog_and_harris_scale_space_detectors_bit_exact(ctx, modsx, oracle, small_pair, det, th):
    """PyramidParams::DetectorType = DET_DOG / DET_HARRIS inside the scale-space loop (ScaleSpaceDetector::Response,
    pyramid.cpp:132-175: dogResponse = the level minus its blur with sigma = norm, HarrisResponse :283-305), with their thresholds
    (pyramid.h:47-67: not squared) and point types (DOG_DARK / DOG_BRIGHT = 10 / 11, HARRIS_* = 30 / 31, pyramid.cpp:92-107): the
    scale-space keypoints with every field, then the affine keypoints, FixedTh and NotLessThanRegions."""
    for img in small_pair[:2]:
        im = ctx.upload(img)
        for mode, regn in ((0, 2000), (4, 120)):
            got = ctx.detect_scalespace(im, modsx.default_hessaff_params(mode=mode, detectorType=det, threshold=th))
            ref = oracle.detect_scalespace(img, oracle.default_params(mode=mode, detectorType=det, threshold=th))
            assert len(ref) > 30, len(ref)
            _check_sskp(got, ref)
            assert set(np.unique(ref["type"])) <= ({10, 11} if det == 1 else {30, 31})
            gk = ctx.detect_affine_keypoints(im, modsx.default_hessaff_params(mode=mode, reg_number=regn, detectorType=det, threshold=th))
            rk = oracle.detect_hessaff(img, oracle.default_params(mode=mode, reg_number=regn, detectorType=det, threshold=th))
            assert len(rk) > 15 and same_records(gk, rk.view(modsx.KEYPOINT))
        im.free()
    with pytest.raises(RuntimeError):
        ctx.detect_scalespace(ctx.upload(small_pair[0]), modsx.default_hessaff_params(detectorType=7))


@pytest.mark.parametrize("mode,regn", [(0, 2000), (4, 150), (2, 40)])
def test_affine_keypoints_bit_exact(ctx, modsx, oracle, small_pair, mode, regn):
    for img in small_pair[:2]:
        im = ctx.upload(img)
        got = ctx.detect_affine_keypoints(im, modsx.default_hessaff_params(mode=mode, reg_number=regn))
        im.free()
        ref = oracle.detect_hessaff(img, oracle.default_params(mode=mode, reg_number=regn))
        assert len(ref) > 20
        assert same_records(got, ref.view(modsx.KEYPOINT))


def test_affine_keypoints_without_baumberg(ctx, modsx, oracle, small_pair):
    """doBaumberg = 0 (HessianAffineParams, detectors/structures.hpp; pyramid / affine detectors skip findAffineShape and export
    the isotropic keypoint): the branch engine.hip takes around the Baumberg stage, in the two export modes that reach it."""
    for mode, regn in ((0, 2000), (4, 150)):
        for img in small_pair[:2]:
            im = ctx.upload(img)
            got = ctx.detect_affine_keypoints(im, modsx.default_hessaff_params(mode=mode, reg_number=regn, doBaumberg=0))
            with_b = ctx.detect_affine_keypoints(im, modsx.default_hessaff_params(mode=mode, reg_number=regn))
            im.free()
            ref = oracle.detect_hessaff(img, oracle.default_params(mode=mode, reg_number=regn, doBaumberg=0))
            assert len(ref) > 20
            assert same_records(got, ref.view(modsx.KEYPOINT))
            assert len(with_b) != len(got) or not same_records(with_b, got)      # the flag is seen


@pytest.mark.parametrize("kw", [dict(smmWindowSize=15), dict(smmWindowSize=11, maxIterations=5), dict(maxIterations=1),
                                dict(maxIterations=7, convergenceThreshold=0.01), dict(convergenceThreshold=0.3)])
def test_baumberg_parameter_variants_bit_exact(ctx, modsx, oracle, small_pair, kw):
    """AffineShape::findAffineShape (affine.cpp:26-169) away from the defaults: other second-moment windows (the
    single-keypoint kernel), iteration caps that cut adaptation short and looser / tighter convergence (the grouped kernel's
    lock-step exits)."""
    img = small_pair[0]
    im = ctx.upload(img)
    got = ctx.detect_affine_keypoints(im, modsx.default_hessaff_params(**kw))
    im.free()
    ref = oracle.detect_hessaff(img, oracle.default_params(**kw))
    assert same_records(got, ref.view(modsx.KEYPOINT))
    if kw.get("maxIterations", 16) > 1:
        assert len(ref) > 20


def test_orientation_and_description_bit_exact(ctx, modsx, oracle, small_pair):
    img = small_pair[0]
    im = ctx.upload(img)
    k = oracle.detect_hessaff(img, oracle.default_params())
    regs = oracle.detect_affine_regions(k)
    for mr, max_ang in ((1.0, 1), (5.1962, 5)):
        ref = oracle.detect_orientation(img, regs, mr_size=mr, max_ang=max_ang)
        got = ctx.detect_orientation(im, regs.view(modsx.REGION), mr_size=mr, max_ang=max_ang)
        assert len(ref) > 20 and same_records(got, ref.view(modsx.REGION))
    ro = oracle.detect_orientation(img, regs)
    rr = oracle.reproject_regions(ro, np.eye(3), img.shape[1], img.shape[0])
    for rootsift in (1, 0, 3, 2):      # RootSIFT, SIFT, HalfRootSIFT, HalfSIFT (siftdesc.cpp:399-442)
        for photo in (1, 0):
            ref = oracle.describe_regions(img, rr, rootsift=rootsift, photo_norm=photo)
            got = ctx.describe_regions(im, rr.view(modsx.REGION), desc_type=rootsift, photo_norm=photo)
            assert np.array_equal(got, ref), (rootsift, photo, int((got != ref).any(1).sum()), len(ref))
            if rootsift >= 2:
                assert not got[:, 64:].any() and got[:, :64].any()
    # large windows: row tiles of a few rows in the fused sampling kernel (blur kernels of 33..170 taps), a window whose row
    # tile does not fit LDS at all (s = 200: 2083 px wide, 459 taps -> k_patch_sample + the global-memory filter), and a
    # small one that is column-filtered in the sampling kernel too
    big = rr[:9].copy()
    for side in ("det_kp", "reproj_kp"):
        big[side]["s"] = [14.0, 20.0, 33.0, 41.5, 56.0, 60.0, 75.0, 7.0, 200.0]
    ref = oracle.describe_regions(img, big)
    got = ctx.describe_regions(im, big.view(modsx.REGION))
    assert np.array_equal(got, ref), int((got != ref).any(1).sum())
    # fast extraction branch (synth-detection.hpp:232-253) and a tiny-scale region (direct branch, i2p <= 0.4)
    ref = oracle.describe_regions(img, rr, fast=1)
    got = ctx.describe_regions(im, rr.view(modsx.REGION), fast=1)
    assert np.array_equal(got, ref)
    tiny = rr[:8].copy()
    tiny["det_kp"]["s"] = 0.9
    assert np.array_equal(ctx.describe_regions(im, tiny.view(modsx.REGION)), oracle.describe_regions(img, tiny))
    im.free()


@pytest.mark.parametrize("mr,max_ang", [(1.0, 1), (1.0, 5), (5.1962, 1), (5.1962, 5), (5.1962, 18)])
def test_orientation_half_mode_bit_exact(ctx, modsx, oracle, small_pair, mr, max_ang):
    """DetectOrientation(..., doHalfSIFT = true, ...) -- what the reference runs for every step whose descriptor list holds a
    Half type (imagerepresentation.cpp:1259-1264): opposite bins of the smoothed 36-bin histogram are folded before the peak
    search (synth-detection.cpp:801-808).  Both measurement regions of the shipped configs x maxAngles 1 / 5 / 18 (= every peak a 36-bin histogram can have); the folded
    and the plain mode must also differ, or the test would not see the flag."""
    differs = 0
    for img in small_pair[:2]:
        im = ctx.upload(img)
        k = oracle.detect_hessaff(img, oracle.default_params())
        regs = oracle.detect_affine_regions(k)
        ref = oracle.detect_orientation(img, regs, mr_size=mr, half=1, max_ang=max_ang)
        got = ctx.detect_orientation(im, regs.view(modsx.REGION), mr_size=mr, half=1, max_ang=max_ang)
        plain = ctx.detect_orientation(im, regs.view(modsx.REGION), mr_size=mr, half=0, max_ang=max_ang)
        im.free()
        assert len(ref) > 20 and same_records(got, ref.view(modsx.REGION))
        differs += int(len(plain) != len(got) or not same_records(plain, got))
    assert differs > 0


@pytest.mark.parametrize("mr,max_ang,half", [(1.0, 1, 0), (5.1962, 5, 0), (5.1962, 5, 1), (5.1962, 18, 0)])
def test_orientation_add_upright_bit_exact(ctx, modsx, oracle, small_pair, mr, max_ang, half):
    """DetectOrientation(..., addUpRight = true) (imagerepresentation.cpp:1265-1269, synth-detection.cpp DetectOrientation): the
    unrotated region is kept beside the oriented ones.  Plain and Half-folded histograms, one and several peaks."""
    img = small_pair[0]
    im = ctx.upload(img)
    k = oracle.detect_hessaff(img, oracle.default_params())
    regs = oracle.detect_affine_regions(k)
    ref = oracle.detect_orientation(img, regs, mr_size=mr, half=half, max_ang=max_ang, upright=1)
    got = ctx.detect_orientation(im, regs.view(modsx.REGION), mr_size=mr, half=half, max_ang=max_ang, upright=1)
    without = ctx.detect_orientation(im, regs.view(modsx.REGION), mr_size=mr, half=half, max_ang=max_ang, upright=0)
    im.free()
    assert len(ref) > len(regs) and same_records(got, ref.view(modsx.REGION))
    assert len(without) < len(got)


def test_step_descriptor_list_one_pass_equals_separate_calls(ctx, modsx, oracle, small_pair):
    """A step with Descriptors = RootSIFT, HalfRootSIFT (iters_mods_cviu_wxbs.ini:35): the view loop orients once with the
    Half-folded histogram and emits both descriptors from one pass over the patches; the first class it returns must be what
    DescribeRegions gives on the Half-oriented list, for either order of the list and against the oracle-side loop."""
    img = small_pair[0]
    im = ctx.upload(img)
    vo = oracle.set_vs_pars([1.0], [1, 2], 360.0, 0.2, 1, [])
    vm = modsx.set_vs_pars([1.0], [1, 2], 360.0, 0.2, 1, [])
    r_ref, d_ref = oracle.detect_describe_views(img, vo, ori=(5.1962, 41, 5, 0.8), descs=[1, 3])
    for order in ([(1, 0.8), (3, 0.8)], [(3, 0.8), (1, 0.8)]):
        par = modsx.default_pair_params(ori_mrSize=5.1962, ori_maxAngles=5, descs=order)
        regs, desc = ctx.detect_describe_views(im, vm, par)
        assert len(r_ref) > 100 and same_records(regs, r_ref.view(modsx.REGION))
        assert np.array_equal(desc, d_ref[0] if order[0][0] == 1 else d_ref[1])
    # a RootSIFT-only step on the same views orients WITHOUT the fold: other regions
    par = modsx.default_pair_params(ori_mrSize=5.1962, ori_maxAngles=5)
    regs1, _ = ctx.detect_describe_views(im, vm, par)
    r1_ref, _ = oracle.detect_describe_views(img, vo, ori=(5.1962, 41, 5, 0.8))
    assert same_records(regs1, r1_ref.view(modsx.REGION)) and len(regs1) != len(r_ref)
    im.free()


def _check_tents(a, b):
    assert len(a) == len(b)
    for f in a.dtype.names:
        assert np.array_equal(a[f], b[f]), f


def test_match_fginn_bit_exact_on_real_descriptors(ctx, oracle, small_pair):
    a, b, _ = small_pair
    _, r1, d1 = oracle_features(oracle, a)
    _, r2, d2 = oracle_features(oracle, b)
    pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
    for ratio, cd in ((0.8, 30.0), (0.9, 10.0), (0.6, 3.0)):
        ref = oracle.match_fginn(d1, d2, pos2, ratio, cd)
        got = ctx.match_fginn(d1, d2, pos2, ratio, cd)
        assert len(ref) > 5
        _check_tents(got, ref)


def test_match_fginn_ties_ranks_and_ragged_sizes(ctx, oracle):
    rs = np.random.RandomState(9)
    for n1, n2 in ((1, 50), (33, 95), (70, 257), (5, 64)):
        # low-entropy descriptors: many exact distance ties, duplicates, zero distances
        d1 = rs.randint(0, 3, (n1, 128)).astype(np.float32) * 40
        d2 = rs.randint(0, 3, (n2, 128)).astype(np.float32) * 40
        d2[n2 // 2:] = d2[: n2 - n2 // 2]                       # duplicated trains
        d1[0] = d2[3]                                          # exact hit (d0 = 0)
        pos2 = rs.uniform(0, 60, (n2, 2))                      # dense positions: long walks through consistent NNs
        for ratio, cd in ((0.8, 30.0), (0.95, 80.0), (0.8, 5.0)):
            _check_tents(ctx.match_fginn(d1, d2, pos2, ratio, cd), oracle.match_fginn(d1, d2, pos2, ratio, cd))
    assert len(ctx.match_fginn(np.zeros((0, 128)), np.zeros((4, 128)), np.zeros((4, 2)))) == 0
    # outside the domain the int8 path is exact on (integers 0..255), and for the branches that are not built, the call
    # fails loudly instead of returning something else than the reference would
    d = rs.randint(0, 255, (8, 128)).astype(np.float32)
    p2 = rs.uniform(0, 60, (8, 2))
    for bad in (d + 0.5, d - 300.0, np.where(np.arange(128) == 3, np.nan, d)):
        with pytest.raises(RuntimeError):
            ctx.match_fginn(bad.astype(np.float32), d, p2)
    with pytest.raises(RuntimeError):
        ctx.match_fginn(d, d, p2, nn=300)             # the walk handles nn up to 256 (the reference's default is 50)


def test_match_fginn_all_points_mode(ctx, oracle, small_pair):
    """ratio >= 1: the "to get all points (for example, for calculating PDF)" branch of MatchFlannFGINN (matching.cpp:397-428) -- a
    record per query, closed by its first contradictive neighbour or by neighbour nn - 1.  Real descriptors, and runs of
    near-duplicates at one place whose walks go down to rank nn - 1 (k_match_pdf: the query's exact sorted list)."""
    a, b, _ = small_pair
    _, r1, d1 = oracle_features(oracle, a)
    _, r2, d2 = oracle_features(oracle, b)
    pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
    for ratio, cd, nn in ((1.0, 30.0, 50), (1.3, 5.0, 20), (1.0, 1000.0, 12)):
        ref = oracle.match_fginn(d1, d2, pos2, ratio, cd, nn)
        assert len(ref) == len(d1)
        _check_tents(ctx.match_fginn(d1, d2, pos2, ratio, cd, nn), ref)
    rs = np.random.RandomState(5)
    n1, n2 = 300, 3000
    e2 = rs.randint(1, 90, (n2, 128)).astype(np.float32)
    e1 = rs.randint(1, 90, (n1, 128)).astype(np.float32)
    p2 = rs.uniform(0, 300, (n2, 2))
    for q in range(0, n1, 2):
        k = int(rs.randint(2, 60))
        start = int(rs.randint(0, n2 - k))
        e2[start:start + k] = np.clip(e1[q][None, :] + rs.randint(-2, 3, (k, 128)), 1, 255)
        p2[start:start + k] = p2[start] + rs.uniform(-3, 3, (k, 2))
    e2[7] = e2[8]                                                   # an exact tie
    for ratio, cd, nn in ((1.0, 30.0, 50), (1.5, 10.0, 8), (1.0, 500.0, 100)):
        _check_tents(ctx.match_fginn(e1, e2, p2, ratio, cd, nn), oracle.match_fginn(e1, e2, p2, ratio, cd, nn))
    # fewer than nn trains: a walk that never meets a contradictive neighbour runs off the list and leaves no record
    _check_tents(ctx.match_fginn(e1[:40], e2[:30], p2[:30], 1.0, 1e6, 50), oracle.match_fginn(e1[:40], e2[:30], p2[:30], 1.0, 1e6, 50))


def test_match_fginn_clustered_near_duplicates_and_split_ranges(ctx, oracle):
    """The paths of the device matcher that small real pairs do not reach: several splits of the train range (n2 > 768),
    index chunks of 12 tiles, runs of up to 40 near-duplicate trains of one query packed into ONE (split, lane-half) stream
    (more than the 16 event slots of a stream with fewer than nn groups in total -> the exact rescan of that stream), exact
    distance ties across tiles and lane halves, walks that end at rank nn - 1 / nn, and a query block boundary (n1 > 256)."""
    rs = np.random.RandomState(31)
    for n1, n2, kmax in ((300, 2100, 40), (40, 5000, 25), (513, 1000, 12), (60, 3000, 150)):   # (the last: runs beyond 64, for nn > 64)
        base = rs.randint(0, 90, (n2, 128)).astype(np.float32)
        d2 = base.copy()
        d1 = rs.randint(0, 90, (n1, 128)).astype(np.float32)
        pos2 = rs.uniform(0, 2000, (n2, 2))
        for q in range(0, n1, 3):
            k = int(rs.randint(2, kmax))
            start = int(rs.randint(0, n2 - k))
            near = np.clip(d1[q][None, :] + rs.randint(-2, 3, (k, 128)), 0, 255)
            d2[start:start + k] = near                     # contiguous: same tiles, same stream
            pos2[start:start + k] = pos2[start] + rs.uniform(-3, 3, (k, 2))
            if q % 6 == 0:
                d2[start + k - 1] = d2[start]               # an exact tie inside the run
        # one near-duplicate per tile, always in the same lane half, over 20 consecutive tiles of the first split: 20 event
        # groups in ONE stream (> 16 slots) but fewer than nn in total -> the stream is rescanned exactly
        for q in (1, 4, 7):
            t0 = 32 * (q % 3)
            for j in range(20):
                t = t0 + 32 * j + 1
                if t < n2:
                    d2[t] = np.clip(d1[q] + rs.randint(-2, 3, 128), 0, 255)
                    pos2[t] = pos2[t0 + 1] + rs.uniform(-3, 3, 2)
        for ratio, cd, nn in ((0.8, 30.0, 50), (0.9, 30.0, 20), (0.8, 2.0, 50), (0.9, 30.0, 100), (0.95, 40.0, 256), (0.9, 30.0, 65)):
            _check_tents(ctx.match_fginn(d1, d2, pos2, ratio, cd, nn), oracle.match_fginn(d1, d2, pos2, ratio, cd, nn))


def test_pair_end_to_end_identical_inliers(ctx, modsx, oracle, small_pair):
    a, b, H = small_pair
    need_ref(oracle)
    ia, ib = ctx.upload(a), ctx.upload(b)
    for seed in (1, 77):
        got = ctx.match_pair(ia, ib, modsx.default_pair_params(ransac_seed=seed))
        ref = oracle_pair(oracle, a, b, seed=seed)
        assert got["n_regions"] == (len(ref["d1"]), len(ref["d2"]))
        assert got["n_tentatives"] == len(ref["tent"]) and got["n_unique"] == len(ref["uniq"])
        _check_tents(got["tentatives"], ref["uniq"])
        rr = ref["ransac"]
        assert np.array_equal(got["ransac_inlier"], rr["inl"])          # identical inlier indices
        assert np.array_equal(got["verified"], rr["keep"]) and got["n_verified"] == rr["n"]
        assert np.abs(normH(got["H"]) - normH(rr["H"])).max() < 1e-4    # H within 1e-4 (north star)
        assert got["ransac_samples"] == rr["samples"]
    ia.free(); ib.free()


@pytest.mark.parametrize("det,th", [(1, 1.5), (2, 400.0)])
def test_pair_end_to_end_with_dog_and_harris_detectors(ctx, modsx, oracle, small_pair, det, th):
    """The whole pair path (detect -> orient -> describe -> match -> duplicate filter -> LO-RANSAC) with DetectorType = DoG / Harris in
    the scale-space loop: the regions' keypoint sub-types, every unique tentative and the inlier set against the oracle's pipeline."""
    a, b, _ = small_pair
    need_ref(oracle)
    ia, ib = ctx.upload(a), ctx.upload(b)
    got = ctx.match_pair(ia, ib, modsx.default_pair_params(ransac_seed=5, detectorType=det, threshold=th))
    ref = oracle_pair(oracle, a, b, seed=5, detectorType=det, threshold=th)
    ia.free(); ib.free()
    assert got["n_regions"] == (len(ref["d1"]), len(ref["d2"])) and len(ref["d1"]) > 50
    assert got["n_tentatives"] == len(ref["tent"]) and got["n_unique"] == len(ref["uniq"]) and len(ref["uniq"]) > 10
    _check_tents(got["tentatives"], ref["uniq"])
    rr = ref["ransac"]
    assert np.array_equal(got["ransac_inlier"], rr["inl"]) and np.array_equal(got["verified"], rr["keep"])


def test_pair_end_to_end_epipolar_verification(ctx, modsx, oracle, small_pair):
    """RANSACPars::useF = 1 (config 5 of BASELINE.json): the same pair verified by exp_ransacFcustom + F_LAF_check.
    The scene is planar, so DEGENSAC's plane-and-parallax branch is what runs."""
    a, b, _ = small_pair
    need_ref(oracle)
    ia, ib = ctx.upload(a), ctx.upload(b)
    got = ctx.match_pair(ia, ib, modsx.default_pair_params(ransac_seed=3, useF=1, LAFCoef=2.0, err_threshold=4.0))
    ia.free(); ib.free()
    ref = oracle_pair(oracle, a, b, seed=3)
    _check_tents(got["tentatives"], ref["uniq"])
    tu = ref["uniq"]
    rr = oracle.loransac_f(ref["pts"], laf_of(ref["r1"], tu["q"]), laf_of(ref["r2"], tu["t0"]), err_threshold=4.0,
                           laf_coef=2.0, seed=3)
    assert np.array_equal(got["ransac_inlier"], rr["inl"]) and np.array_equal(got["verified"], rr["keep"])
    assert got["n_verified"] == rr["n"] and got["ransac_samples"] == rr["samples"] and rr["n"] > 50
    Fa, Fb = rr["F"] / np.linalg.norm(rr["F"]), got["H"] / np.linalg.norm(got["H"])
    if (Fa * Fb).sum() < 0:
        Fb = -Fb
    assert np.abs(Fa - Fb).max() < 1e-6


def test_device_side_detection_order_opt_in_is_bit_exact(oracle, small_pair):
    """MODSX_DEVICE_ORDER=1 (kernels_cand.hip: device radix sort of the candidates + first-visitor claim of the converged pixel +
    ordered compaction; off by default because it measured slower than the host's sort): the same keypoints as the oracle, in the
    same order, for one image and for a multi-image launch set.  Runs in a child process (the switch is read once per process)."""
    import json, os, subprocess, sys, tempfile
    code = (
        "import sys, json, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import mods_amd\n"
        "from mods_amd import synthetic\n"
        "from oracle import pyoracle as O\n"
        "from common import same_records\n"
        "a, b, _ = synthetic.make_pair(rows=240, cols=320, nblobs=420, seed=777)\n"
        "ctx = mods_amd.Context(0)\n"
        "ok = True\n"
        "for img in (a, b, a):\n"
        "    im = ctx.upload(img)\n"
        "    for mode in (0, 4):\n"
        "        ss = ctx.detect_scalespace(im, mods_amd.default_hessaff_params(mode=mode))\n"
        "        ok = ok and same_records(ss, O.detect_scalespace(img, O.default_params(mode=mode)).view(mods_amd.SSKP))\n"
        "        k = ctx.detect_affine_keypoints(im, mods_amd.default_hessaff_params(mode=mode))\n"
        "        ok = ok and same_records(k, O.detect_hessaff(img, O.default_params(mode=mode)).view(mods_amd.KEYPOINT))\n"
        "    im.free()\n"
        "views = mods_amd.set_vs_pars([1.0], [1, 2, 3], 360.0, 0.2, 1, [])\n"
        "vo = O.set_vs_pars([1.0], [1, 2, 3], 360.0, 0.2, 1, [])\n"
        "im = ctx.upload(a)\n"
        "r, d = ctx.detect_describe_views(im, views, mods_amd.default_pair_params())\n"
        "rr, dd = O.detect_describe_views(a, vo)\n"
        "ok = ok and same_records(r, rr.view(mods_amd.REGION)) and bool(np.array_equal(d, dd))\n"
        "print(json.dumps({'ok': bool(ok), 'n': int(len(r))}))\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, MODSX_DEVICE_ORDER="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["n"] > 300


def test_cat_pair_golden_counts(ctx, modsx, cat_pair):
    cat, cat2, _ = cat_pair
    i1, i2 = ctx.upload(cat), ctx.upload(cat2)
    p = modsx.default_hessaff_params()
    assert len(ctx.detect_affine_keypoints(i1, p)) == 255
    assert len(ctx.detect_affine_keypoints(i2, p)) == 414
    p = modsx.default_hessaff_params(mode=4)
    assert len(ctx.detect_affine_keypoints(i1, p)) == 2000
    assert len(ctx.detect_affine_keypoints(i2, p)) == 2000
    i1.free(); i2.free()


def test_cat_pair_every_stage_field_by_field(ctx, modsx, oracle, cat_pair):
    """configs[0] on the one input whose grey values are not integers: cat.png / cat2.png are colour images, (B + G + R) / 3
    leaves thirds (synth-detection.cpp:256-262), so every f32 rounding of the blur, the sampling and the Hessian actually
    bites here (the synthetic images are floor'ed).  Every stage against the oracle, field by field: scale-space keypoints,
    affine keypoints (FixedTh 255 / 414 and NotLessThanRegions 2000 / 2000), oriented regions, RootSIFT descriptors, the
    tentatives of the pair; then the 11-view HessianAffine step of iters_mods_cviu.ini:56-62 (regions + descriptors of every
    view of both images) and the pair through it."""
    cat, cat2, _ = cat_pair
    feats = []
    for bgr in (cat, cat2):
        g = oracle.gray_from_bgr(bgr)
        assert np.abs(g * 3 - np.round(g * 3)).max() < 1e-3 and (np.abs(g - np.round(g)) > 0.2).any()      # thirds, not integers
        im = ctx.upload(bgr)
        ss_ref = oracle.detect_scalespace(g, oracle.default_params())
        ss_got = ctx.detect_scalespace(im, modsx.default_hessaff_params())
        assert len(ss_ref) > 200 and same_records(ss_got, ss_ref.view(modsx.SSKP))
        for mode, n in ((0, None), (4, 2000)):
            k_ref = oracle.detect_hessaff(g, oracle.default_params(mode=mode))
            k_got = ctx.detect_affine_keypoints(im, modsx.default_hessaff_params(mode=mode))
            assert same_records(k_got, k_ref.view(modsx.KEYPOINT)) and (n is None or len(k_ref) == n)
        r = oracle.detect_affine_regions(k_ref)                       # the 2000 strongest
        for mr, max_ang, half in ((1.0, 1, 0), (5.1962, 5, 1)):
            ro = oracle.detect_orientation(g, r, mr_size=mr, max_ang=max_ang, half=half)
            go = ctx.detect_orientation(im, r.view(modsx.REGION), mr_size=mr, max_ang=max_ang, half=half)
            assert len(ro) > 1500 and same_records(go, ro.view(modsx.REGION))
        ro = oracle.detect_orientation(g, r)
        rr = oracle.reproject_regions(ro, np.eye(3), g.shape[1], g.shape[0])
        for t in (1, 3):
            d_ref = oracle.describe_regions(g, rr, rootsift=t)
            d_got = ctx.describe_regions(im, rr.view(modsx.REGION), desc_type=t)
            assert np.array_equal(d_got, d_ref), (t, int((d_got != d_ref).any(1).sum()))
        feats.append((g, im, rr, oracle.describe_regions(g, rr)))
    (g1, i1, r1, d1), (g2, i2, r2, d2) = feats
    pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
    _check_tents(ctx.match_fginn(d1, d2, pos2, 0.8, 30.0), oracle.match_fginn(d1, d2, pos2, 0.8, 30.0))
    # the 11 views of [HessianAffine4]: TiltSet 1,2,4,6,8, Phi 360, initSigma 0.2
    vo = oracle.set_vs_pars([1.0], [1, 2, 4, 6, 8], 360.0, 0.2, 1, [])
    vm = modsx.set_vs_pars([1.0], [1, 2, 4, 6, 8], 360.0, 0.2, 1, [])
    par = modsx.default_pair_params(ransac_seed=3)
    import os
    acc = []
    for g, im in ((g1, i1), (g2, i2)):
        r_ref, d_ref = oracle.detect_describe_views(g, vo, threads=min(16, os.cpu_count() or 1))
        r_got, d_got = ctx.detect_describe_views(im, vm, par)
        assert len(r_ref) > 400 and same_records(r_got, r_ref.view(modsx.REGION)) and np.array_equal(d_got, d_ref)
        acc.append((r_ref, d_ref))
    (r1, d1), (r2, d2) = acc
    pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
    tent = oracle.match_fginn(d1, d2, pos2, 0.8, 30.0)
    pts = np.stack([r1["reproj_kp"]["x"][tent["q"]], r1["reproj_kp"]["y"][tent["q"]],
                    r2["reproj_kp"]["x"][tent["t0"]], r2["reproj_kp"]["y"][tent["t0"]]], 1)
    order, keep = oracle.duplicate_filtering(pts, tent["ratio"], 2.0, True)
    got = ctx.match_pair_views(i1, i2, vm, par)
    assert got["n_tentatives"] == len(tent)
    _check_tents(got["tentatives"], tent[order[keep]])
    need_ref(oracle)
    tu, pu = tent[order[keep]], pts[order[keep]]
    rr = oracle.loransac_h(pu, laf_of(r1, tu["q"]), laf_of(r2, tu["t0"]), seed=3)
    assert np.array_equal(got["ransac_inlier"], rr["inl"]) and np.array_equal(got["verified"], rr["keep"])
    i1.free(); i2.free()


def test_full_size_pair_properties(ctx, modsx, oracle):
    """BASELINE config 1: one 1024x768 synthetic pair, 1 view.  Checked through size-independent properties
    plus a full oracle comparison of the (cheap) detection stage."""
    from mods_amd import synthetic
    a, b, H = synthetic.make_pair()
    ia, ib = ctx.upload(a), ctx.upload(b)
    got = ctx.match_pair(ia, ib, modsx.default_pair_params(ransac_seed=5))
    assert got["n_regions"][0] > 1500 and got["n_tentatives"] > 500
    assert got["n_verified"] > 300
    assert np.abs(normH(got["H"]) - H).max() < 1.0
    t = got["tentatives"]
    assert np.all(t["ratio"] <= 0.8 + 1e-12) and np.all(np.diff(np.abs(t["ratio"])) >= 0)   # sorted by FGINN ratio
    assert np.all(t["d1"] <= t["d2by2ndcl"]) and np.all(t["d2by2ndcl"] <= t["d2"])
    k = ctx.detect_affine_keypoints(ia, modsx.default_hessaff_params())
    assert same_records(k, oracle.detect_hessaff(a, oracle.default_params()).view(modsx.KEYPOINT))
    again = ctx.match_pair(ia, ib, modsx.default_pair_params(ransac_seed=5))                # idempotent
    assert np.array_equal(again["ransac_inlier"], got["ransac_inlier"]) and np.array_equal(again["H"], got["H"])
    ia.free(); ib.free()


def test_grouped_pairs_equal_single_pairs(modsx, small_pair):
    """modsx_match_pairs runs up to four pairs (eight images, mixed sizes) as one batch per context; every result
    must be the one modsx_match_pair returns for that pair alone."""
    from mods_amd import synthetic
    a, b, _ = small_pair
    a2, b2, _ = synthetic.make_pair(rows=200, cols=272, nblobs=260, seed=31)
    a3, b3, _ = synthetic.make_pair(rows=256, cols=256, nblobs=300, seed=32)
    ctxs = [modsx.Context(0), modsx.Context(0)]
    blank = np.full((96, 128), 90, np.float32)   # no keypoints: an empty problem inside a batched match launch
    hosts = [(a, b), (blank, b2), (a2, b2), (a3, b3), (b, a), (a2, b3[:200, :]), (a3, blank), (a, b), (b2, a2), (blank, blank)]
    dev = [(ctxs[0].upload(x), ctxs[0].upload(y)) for x, y in hosts]
    par = modsx.default_pair_params(ransac_seed=9)
    singles = [ctxs[0].match_pair(x, y, par) for x, y in dev]
    for nctx in (1, 2):
        got = modsx.match_pairs(ctxs[:nctx], [x for x, _ in dev], [y for _, y in dev], par)
        assert len(got) == len(singles)
        for g, s in zip(got, singles):
            assert g["n_regions"] == s["n_regions"] and g["n_tentatives"] == s["n_tentatives"]
            assert g["n_verified"] == s["n_verified"] and g["ransac_samples"] == s["ransac_samples"]
            for f in s["tentatives"].dtype.names:
                assert np.array_equal(g["tentatives"][f], s["tentatives"][f]), f
            assert np.array_equal(g["verified"], s["verified"]) and np.array_equal(g["H"], s["H"])
    for x, y in dev:
        x.free(); y.free()
    for c in ctxs:
        c.close()


def test_degenerate_inputs_do_not_break_the_path(ctx, modsx, oracle):
    """Empty and ragged inputs: blank images (no keypoints), images smaller than the detector border, one-sided
    emptiness, zero pairs, a 1-row image for MSER; every call returns the reference's (empty) result, none raises."""
    blank = np.full((96, 128), 90, np.float32)
    tiny = np.random.RandomState(1).uniform(0, 255, (9, 11)).astype(np.float32)
    from mods_amd import synthetic
    a, b, _ = synthetic.make_pair(rows=120, cols=160, nblobs=120, seed=5)
    par = modsx.default_pair_params(ransac_seed=2)
    ims = {k: ctx.upload(v) for k, v in dict(blank=blank, tiny=tiny, a=a, b=b).items()}
    for x, y in (("blank", "blank"), ("blank", "a"), ("a", "blank"), ("tiny", "tiny"), ("tiny", "a")):
        r = ctx.match_pair(ims[x], ims[y], par)
        assert r["n_verified"] == 0 and r["n_tentatives"] == 0 and len(r["tentatives"]) == 0
        assert r["n_regions"][0] == len(oracle_features(oracle, dict(blank=blank, tiny=tiny, a=a, b=b)[x])[2])
    assert modsx.match_pairs([ctx], [], [], par) == []
    assert len(ctx.detect_affine_keypoints(ims["blank"], modsx.default_hessaff_params())) == 0
    assert len(ctx.detect_msers(ims["blank"])) == 0 and len(ctx.detect_msers(ims["tiny"])) == len(oracle.detect_msers(tiny))
    views = modsx.set_vs_pars([1.0], [1, 2], 360.0, 0.2, 1, [])
    regs, desc = ctx.detect_describe_views(ims["blank"], views, par)
    assert len(regs) == 0 and desc.shape == (0, 128)
    got, done = ctx.match_ladder(ims["blank"], ims["a"], [(views, 0.8)], par, min_matches=10)
    assert done == 1 and got["n_verified"] == 0
    for im in ims.values():
        im.free()
    # beyond the samplers' 32-bit pixel addressing: refused, not mis-sampled
    with pytest.raises(RuntimeError):
        ctx.upload(np.zeros((2, 16385), np.uint8))
    with pytest.raises(RuntimeError):
        ctx.wrap_device(1 << 20, 16384, 8192)


def test_full_hd_pair_runs(ctx, modsx):
    """configs[4] geometry (1920x1080): sizes that are not multiples of any tile, > 10 k regions per image."""
    from mods_amd import synthetic
    a, b, H = synthetic.make_pair(rows=1079, cols=1919, nblobs=11000, seed=21)
    ia, ib = ctx.upload(a), ctx.upload(b)
    r = ctx.match_pair(ia, ib, modsx.default_pair_params(ransac_seed=4))
    ia.free(); ib.free()
    assert min(r["n_regions"]) > 4000 and r["n_verified"] > 500
    assert np.abs(normH(r["H"]) - H).max() < 1.0


WXBS_DESCS = [(1, 0.8), (3, 0.8)]    # Descriptors=RootSIFT,HalfRootSIFT; FGINNThreshold=0.8,0.8 ([HessianAffine4], iters_mods_cviu_wxbs.ini:61-62)


def _wxbs_params(modsx, seed, useF, descs=WXBS_DESCS):
    """config_iter_mods_cviu_wxbs.ini: [HessianAffine] :13-27, [DominantOrientation] :102-108, [SIFTDescriptor] :109-118 with the
    RootSIFT + HalfRootSIFT classes of iters_mods_cviu_wxbs.ini:35,48,61, [Matching] contradDist :173, [DuplicateFiltering]
    :179-182, [RANSAC] :185-194."""
    return modsx.default_pair_params(
        mode=4, threshold=5.3333, reg_number=2000,                 # NotLessThanRegions / 2000
        ori_mrSize=5.1962, ori_maxAngles=5, ori_threshold=0.8,
        desc_mrSize=5.1962, desc_photoNorm=1, desc_maxBinValue=0.2, descs=descs,
        contradDist=10.0, duplicateDist=3.0,
        err_threshold=4.0, confidence=0.99, max_samples=1000000, localOptimization=1, LAFCoef=3.0, HLAFCoef=13.0,
        doSymmCheck=1, useF=useF, ransac_seed=seed)


def _wxbs_oracle(O, a, b, descs=WXBS_DESCS):
    """One identity-view step of the WxBS ladder as the reference runs it (imagerepresentation.cpp:693-706, 1259-1264,
    1288-1296; correspondencebank.cpp:291-347, 117-179): ONE oriented list per image -- doHalfSIFT = true because the step
    carries HalfRootSIFT --, both descriptors on it, each class matched with its threshold, tentatives concatenated in
    descriptor-name order (HalfRootSIFT before RootSIFT) with indices into the concatenated region lists."""
    p = O.default_params(mode=4, threshold=5.3333, reg_number=2000)
    half = 1 if any(t >= 2 for t, _ in descs) else 0
    feats = []
    for g in (a, b):
        k = O.detect_hessaff(g, p)
        r = O.detect_affine_regions(k)
        ro = O.detect_orientation(g, r, mr_size=5.1962, half=half, max_ang=5, th=0.8)
        rr = O.reproject_regions(ro, np.eye(3), g.shape[1], g.shape[0])
        feats.append((rr, {t: O.describe_regions(g, rr, mr_size=5.1962, rootsift=t) for t, _ in descs}))
    (r1, d1), (r2, d2) = feats
    pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
    T, o1, o2 = [], 0, 0
    for t in (3, 2, 1, 0):
        thr = dict(descs).get(t)
        if thr is None:
            continue
        tt = O.match_fginn(d1[t], d2[t], pos2, thr, 10.0).copy()
        tt["q"] += o1
        for f in ("t0", "t1", "tj"):
            tt[f] = np.where(tt[f] >= 0, tt[f] + o2, tt[f])
        T.append(tt)
        o1 += len(r1); o2 += len(r2)
    tent = np.concatenate(T)
    R1, R2 = np.concatenate([r1] * len(descs)), np.concatenate([r2] * len(descs))
    pts = np.stack([R1["reproj_kp"]["x"][tent["q"]], R1["reproj_kp"]["y"][tent["q"]],
                    R2["reproj_kp"]["x"][tent["t0"]], R2["reproj_kp"]["y"][tent["t0"]]], 1)
    order, keep = O.duplicate_filtering(pts, tent["ratio"], 3.0, True)
    sel = order[keep]
    return R1, R2, tent[sel], pts[sel], len(tent)


def test_wxbs_config_full_hd_pair_h_and_f(ctx, modsx, oracle):
    """configs[4] of BASELINE.json end to end on one 1920x1080 pair with the WxBS parameter set (NotLessThanRegions 2000,
    maxAngles 5 on the 5.1962 measurement region with the Half-folded orientation histogram, the RootSIFT and the HalfRootSIFT
    class matched separately, contradDist 10, duplicateDist 3, err_threshold 4, max_samples 1e6) against the CPU oracle: region
    counts, the de-duplicated tentative list field by field, and for both verification types (H: exp_ransacHcustom +
    H_LAF_check 13, F: exp_ransacFcustom + DEGENSAC + F_LAF_check 3) the RANSAC inlier flags, the verified set and the model."""
    from mods_amd import synthetic
    need_ref(oracle)
    a, b, H = synthetic.make_pair(rows=1080, cols=1920, nblobs=3000, seed=77)
    r1, r2, tu, pu, ntent = _wxbs_oracle(oracle, a, b)
    ia, ib = ctx.upload(a), ctx.upload(b)
    for useF in (0, 1):
        got = ctx.match_pair(ia, ib, _wxbs_params(modsx, 5, useF))
        assert got["n_regions"] == (len(r1), len(r2)) and min(got["n_regions"]) >= 4000     # two classes x >= 2000 regions
        assert got["n_tentatives"] == ntent
        _check_tents(got["tentatives"], tu)
        assert (tu["q"] >= len(r1) // 2).any() and (tu["q"] < len(r1) // 2).any()           # both classes survive the filter
        if useF:
            rr = oracle.loransac_f(pu, laf_of(r1, tu["q"]), laf_of(r2, tu["t0"]), err_threshold=4.0, max_samples=1000000,
                                   laf_coef=3.0, seed=5)
            Fa, Fb = rr["F"] / np.linalg.norm(rr["F"]), got["H"] / np.linalg.norm(got["H"])
            if (Fa * Fb).sum() < 0:
                Fb = -Fb
            assert np.abs(Fa - Fb).max() < 1e-6
        else:
            rr = oracle.loransac_h(pu, laf_of(r1, tu["q"]), laf_of(r2, tu["t0"]), err_threshold=4.0, max_samples=1000000,
                                   hlaf_coef=13.0, seed=5)
            assert np.abs(normH(got["H"]) - normH(rr["H"])).max() < 1e-4
            assert np.abs(normH(got["H"]) - H).max() < 1.0
        assert np.array_equal(got["ransac_inlier"], rr["inl"]) and np.array_equal(got["verified"], rr["keep"])
        assert got["n_verified"] > 200
    ia.free(); ib.free()

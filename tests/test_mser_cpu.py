"""MSER.  PARITY UNPINNED: the reference's MSER sources cannot be compiled here, so the product
(mods_amd/csrc/mser.cpp, array union-find) is checked against the independent restatement that follows the reference's own
data structures (oracle/oracle_mser.cpp), plus properties that hold for any correct MSER."""
import numpy as np
import pytest

from mods_amd import synthetic


def _images():
    rs = np.random.RandomState(0)
    a, b, _ = synthetic.make_pair(rows=240, cols=320, nblobs=420, seed=777)
    ramp = np.clip(np.add.outer(np.arange(100.), np.arange(140.)) + rs.normal(0, 6, (100, 140)), 0, 255)
    steps = (np.add.outer(np.arange(90) // 9, np.arange(130) // 13) * 9 % 256).astype(np.float32)
    return [a, b, synthetic.blob_image(300, 400, 900, 5), rs.uniform(0, 255, (120, 150)).astype(np.float32),
            ramp.astype(np.float32), steps, np.full((40, 50), 77, np.float32), rs.uniform(0, 255, (1, 37)).astype(np.float32),
            rs.uniform(0, 255, (23, 1)).astype(np.float32)]


PARAM_SETS = [dict(), dict(min_margin=5.0, min_size=10, max_area=0.2), dict(mode=2, reg_number=40),
              dict(mode=4, reg_number=300), dict(relative=1, min_margin=3.0), dict(mode=1, rel_threshold=0.5),
              dict(mode=3, rel_reg_number=0.25), dict(min_size=1, min_margin=2.0, max_area=1.0)]


@pytest.mark.parametrize("pi", range(len(PARAM_SETS)))
def test_product_equals_restatement(modsx, oracle, pi):
    kw = PARAM_SETS[pi]
    total = 0
    for img in _images():
        ref = oracle.detect_msers(img, **kw)
        got = modsx.detect_msers_u8(img.astype(np.uint8), modsx.default_mser_params(**kw))
        assert len(got) == len(ref)
        for f in ref.dtype.names:
            assert np.array_equal(got[f], ref[f]), (f, img.shape)
        total += len(ref)
    assert total > 50


def test_tilt_and_zoom_scale_the_region_budget(modsx, oracle):
    img = _images()[0]
    for tilt, zoom in ((3.0, 1.0), (1.0, 0.25), (2.0, 0.5)):
        kw = dict(mode=2, reg_number=60)
        ref = oracle.detect_msers(img, tilt=tilt, zoom=zoom, **kw)
        got = modsx.detect_msers_u8(img.astype(np.uint8), modsx.default_mser_params(**kw), tilt=tilt, zoom=zoom)
        want = 60 if not (tilt > 2.0 or zoom < 0.5) else int(np.floor(zoom * 2.0 * 60 / tilt))
        assert len(got) == len(ref) and len(got) <= want
        for f in ref.dtype.names:
            assert np.array_equal(got[f], ref[f])


def test_polarity_symmetry_and_region_properties(modsx):
    img = _images()[2].astype(np.uint8)
    k = modsx.detect_msers_u8(img)
    ki = modsx.detect_msers_u8(255 - img)
    plus, minus = k[k["sub_type"] == 21], k[k["sub_type"] == 20]
    iplus, iminus = ki[ki["sub_type"] == 21], ki[ki["sub_type"] == 20]
    # MSER- of an image are the MSER+ of its negative (same list, same order) and vice versa
    for f in ("x", "y", "a11", "a12", "a21", "a22", "response"):
        assert np.array_equal(plus[f], iminus[f]) and np.array_equal(minus[f], iplus[f])
    assert len(plus) > 20 and len(minus) > 20
    assert np.all(k["x"] >= 0) and np.all(k["x"] <= img.shape[1]) and np.all(k["y"] >= 0) and np.all(k["y"] <= img.shape[0])
    assert np.all(k["response"] >= 8) and np.all(k["s"] == 1.0)
    det = k["a11"] * k["a22"] - k["a12"] * k["a21"]
    assert np.all(det > 0) and np.allclose(k["a12"], k["a21"])          # sqrt of a covariance: symmetric positive definite
    # area bounds: pi * det(A) * 4 approximates the region area of an ellipse-like region (covariance of a uniform disc
    # of radius r is r^2 / 4): it must respect min_size / max_area up to shape effects
    area = np.pi * 4 * det
    assert area.min() > 5 and area.max() < 4 * 0.05 * img.size


def test_disc_gives_its_moments(modsx):
    img = np.full((120, 160), 200, np.uint8)
    yy, xx = np.mgrid[:120, :160]
    disc = (yy - 60) ** 2 + (xx - 80) ** 2 < 12 ** 2
    img[disc] = 50
    k = modsx.detect_msers_u8(img, modsx.default_mser_params(max_area=0.5))
    assert len(k) == 1 and k["sub_type"][0] == 21 and k["response"][0] == 150
    assert abs(k["x"][0] - (xx[disc].mean() + 0.5)) < 1e-9 and abs(k["y"][0] - (yy[disc].mean() + 0.5)) < 1e-9
    r2 = k["a11"][0] * k["a22"][0] - k["a12"][0] * k["a21"][0]
    assert abs(np.sqrt(r2) * 2 - 12) < 0.5                               # sqrt(cov) of a disc = r / 2


def test_view_set_on_the_host_pool_equals_single_views(modsx):
    """detect_msers_views (the engine's MSER step: 2 n (view, polarity) tasks on the host pool, the two tasks of a view share one bin
    sort and scatter half of its rows each) against the one-view entry point, view by view -- several rounds, so that task order and
    who sorts which half vary."""
    import ctypes as C
    L = modsx.lib()
    imgs = [np.ascontiguousarray(i.astype(np.uint8)) for i in _images()]
    imgs = imgs + [np.ascontiguousarray(imgs[0][::2, ::2]), np.ascontiguousarray(imgs[2][:, ::3])]
    n = len(imgs)
    par = modsx.default_mser_params(min_margin=5.0, min_size=10, max_area=0.2)
    single = [modsx.detect_msers_u8(g, par) for g in imgs]
    assert sum(len(s) for s in single) > 200
    ptrs = (C.c_void_p * n)(*[g.ctypes.data for g in imgs])
    rows = (C.c_int * n)(*[g.shape[0] for g in imgs]); cols = (C.c_int * n)(*[g.shape[1] for g in imgs])
    ones = (C.c_double * n)(*([1.0] * n))
    for rep in range(6):
        counts = (C.c_int * n)()
        out = C.c_void_p()
        tot = L.modsx_debug_msers_views_u8(ptrs, rows, cols, n, C.byref(par), ones, ones, counts, C.byref(out))
        assert tot == sum(len(s) for s in single)
        arr = np.frombuffer((C.c_char * (max(tot, 1) * modsx.KEYPOINT.itemsize)).from_address(out.value), dtype=modsx.KEYPOINT, count=tot).copy()
        L.modsx_free(out)
        at = 0
        for i in range(n):
            assert counts[i] == len(single[i])
            got = arr[at:at + counts[i]]; at += counts[i]
            for f in got.dtype.names:
                assert np.array_equal(got[f], single[i][f]), (rep, i, f)


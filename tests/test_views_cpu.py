"""CPU: view ladder (SetVSPars) and the oracle's view synthesis."""
import numpy as np


def _vt(v):
    return (round(v.zoom, 6), round(v.tilt, 6), round(v.phi, 6), round(v.InitSigma, 6), v.doBlur)


def test_set_vs_pars_counts_and_dedup(modsx, oracle):
    # TiltSet=1,2,3,4,6 Phi=360 -> 1+1+1+2+3 = 8 views; 1,2,4,6,8 -> 11 / 31 / 61
    for tilts, phi, expect in (([1, 2, 3, 4, 6], 360.0, 8), ([1, 2, 4, 6, 8], 360.0, 11), ([1, 2, 4, 6, 8], 120.0, 31),
                               ([1, 2, 4, 6, 8], 60.0, 61)):
        a = modsx.set_vs_pars([1.0], tilts, phi, 0.2, 1, [])
        b = oracle.set_vs_pars([1.0], tilts, phi, 0.2, 1, [])
        assert len(a) == expect and [_vt(v) for v in a] == [_vt(v) for v in b]
    prev_a, prev_b = [], []
    s1a = modsx.set_vs_pars([1.0], [1, 2, 4, 6, 8], 360.0, 0.2, 1, prev_a)
    s1b = oracle.set_vs_pars([1.0], [1, 2, 4, 6, 8], 360.0, 0.2, 1, prev_b)
    s2a = modsx.set_vs_pars([1.0], [1, 2, 4, 6, 8], 120.0, 0.2, 1, prev_a)     # step 5 of iters_mods_cviu.ini
    s2b = oracle.set_vs_pars([1.0], [1, 2, 4, 6, 8], 120.0, 0.2, 1, prev_b)
    assert len(s1a) == 11 and len(s2a) == 20 and [_vt(v) for v in s2a] == [_vt(v) for v in s2b]   # 11 + 20 + 30 = 61
    s3a = modsx.set_vs_pars([1.0], [1, 2, 4, 6, 8], 60.0, 0.2, 1, prev_a)
    assert len(s3a) == 30 and len(prev_a) == 61
    neg = modsx.set_vs_pars([1.0], [2.0], -1.0, 0.5, 1, [])                      # "no rotation" mode: vertical tilt
    assert len(neg) == 2 and neg[0].tilt == -2.0 and neg[1].tilt == 2.0 and neg[1].phi == 0   # :141-170
    empty = modsx.set_vs_pars([], [], 360.0, 0.5, 1, [])
    assert len(empty) == 1 and empty[0].tilt == 0 and empty[0].doBlur == 0


def test_warp_affine_identity_and_shift(oracle):
    rs = np.random.RandomState(0)
    img = rs.uniform(0, 255, (20, 30)).astype(np.float32)
    same = oracle.warp_affine(img, [1, 0, 0, 0, 1, 0], 20, 30)
    assert np.array_equal(same[:-1, :-1], img[:-1, :-1])         # last row/col blend with the 128 border? no: weights 1,0
    sh = oracle.warp_affine(img, [1, 0, 2, 0, 1, 3], 20, 30)      # dst(x,y) = src(x-2, y-3)
    assert np.array_equal(sh[3:-1, 2:-1], img[:-4, :-3]) and np.all(sh[:3] == 128) and np.all(sh[:, :2] == 128)
    half = oracle.warp_affine(img, [0.5, 0, 0, 0, 1, 0], 20, 15)  # tilt 2: dst x -> src 2x (exact sample points)
    assert np.array_equal(half[:-1, :], img[:-1, 0:30:2])


def test_synth_view_geometry(oracle, small_pair):
    a = small_pair[0]
    v = oracle.make_view(tilt=2.0, phi=0.0, init_sigma=0.2)
    img, H, ident = oracle.synth_view(a, v)
    assert not ident and img.shape == (240, 160) and np.allclose(H, [[0.5, 0, 0], [0, 1, 0], [0, 0, 1]])
    v = oracle.make_view(tilt=1.0)
    img, H, ident = oracle.synth_view(a, v)
    assert ident and np.array_equal(img, a) and np.array_equal(H, np.eye(3))
    v = oracle.make_view(tilt=4.0, phi=np.pi / 2, init_sigma=0.2)
    img, H, ident = oracle.synth_view(a, v)
    assert img.shape == (320, 60)
    # a point of the original maps into the view by H
    p = H @ np.array([100.0, 50.0, 1.0])
    assert 0 <= p[0] < img.shape[1] and 0 <= p[1] < img.shape[0]


def test_multiview_oracle_pipeline(oracle, small_pair):
    a = small_pair[0]
    views = oracle.set_vs_pars([1.0], [1, 2, 3], 360.0, 0.2, 1, [])
    regs, desc = oracle.detect_describe_views(a, views)
    assert len(regs) == len(desc) > 150
    ids = regs["img_id"]
    assert set(ids.tolist()) == {0, 1, 2} and np.all(np.diff(ids) >= 0)
    x, y = regs["reproj_kp"]["x"], regs["reproj_kp"]["y"]
    assert x.min() > 0 and x.max() < a.shape[1] and y.min() > 0 and y.max() < a.shape[0]
    v1 = regs[ids == 1]
    assert np.allclose(v1["reproj_kp"]["x"], v1["det_kp"]["x"] * 2.0) and np.allclose(v1["reproj_kp"]["y"], v1["det_kp"]["y"])

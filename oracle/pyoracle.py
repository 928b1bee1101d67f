"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (mods_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KEYPOINT = np.dtype([("x", "f8"), ("y", "f8"), ("a11", "f8"), ("a12", "f8"), ("a21", "f8"), ("a22", "f8"),
                     ("s", "f8"), ("response", "f8"), ("octave_number", "i4"), ("pyramid_scale", "f8"),
                     ("sub_type", "i4")], align=True)
REGION = np.dtype([("img_id", "i4"), ("img_reproj_id", "i4"), ("id", "i4"), ("parent_id", "i4"), ("type", "i4"),
                   ("det_kp", KEYPOINT), ("reproj_kp", KEYPOINT)], align=True)
SSKP = np.dtype([("octave", "i4"), ("level", "i4"), ("r0", "i4"), ("c0", "i4"), ("r", "i4"), ("c", "i4"),
                 ("type", "i4"), ("pad", "i4"), ("b0", "f4"), ("b1", "f4"), ("b2", "f4"), ("val", "f4"),
                 ("x", "f4"), ("y", "f4"), ("s", "f4"), ("pixelDistance", "f4")], align=True)
TENT = np.dtype([("q", "i4"), ("t0", "i4"), ("tj", "i4"), ("t1", "i4"), ("d1", "f8"), ("d2", "f8"),
                 ("d2by2ndcl", "f8"), ("ratio", "f8")], align=True)
assert KEYPOINT.itemsize == 88 and REGION.itemsize == 200 and SSKP.itemsize == 64 and TENT.itemsize == 48


class HessAffParams(C.Structure):
    _fields_ = [("threshold", C.c_float), ("mode", C.c_int), ("reg_number", C.c_int),
                ("rel_threshold", C.c_float), ("rel_reg_number", C.c_float), ("numberOfScales", C.c_int),
                ("initialSigma", C.c_float), ("edgeEigenValueRatio", C.c_double), ("border", C.c_int),
                ("maxIterations", C.c_int), ("convergenceThreshold", C.c_float), ("smmWindowSize", C.c_int),
                ("affInitialSigma", C.c_float), ("doBaumberg", C.c_int), ("detectorType", C.c_int)]


class View(C.Structure):
    _fields_ = [("zoom", C.c_double), ("tilt", C.c_double), ("phi", C.c_double), ("InitSigma", C.c_double),
                ("doBlur", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp", ".h", ".c"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/degensac"):
        ref = os.path.join(_HERE, "_ref", "libdegensac_ref.so")
        if force or not os.path.exists(ref):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_atan2lut.restype = C.c_float
        _lib.orc_atan2lut.argtypes = [C.c_float, C.c_float]
        _lib.orc_atan_lut.restype = C.POINTER(C.c_double)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def default_params(**kw):
    p = HessAffParams()
    lib().orc_default_hessaff_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def gray_from_bgr(bgr):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    out = np.empty(bgr.shape[:2], np.float32)
    lib().orc_gray_from_bgr_u8(_p(bgr), bgr.shape[0], bgr.shape[1], _p(out))
    return out


def gaussian_kernel(n, sigma):
    out = np.empty(n, np.float32)
    lib().orc_gaussian_kernel(C.c_int(n), C.c_double(sigma), _p(out))
    return out


def blur_ksize(sigma):
    return lib().orc_blur_ksize(C.c_float(sigma))


def gaussian_blur(img, sigma):
    img = _f32(img)
    out = np.empty_like(img)
    lib().orc_gaussian_blur(_p(img), img.shape[0], img.shape[1], C.c_float(sigma), _p(out))
    return out


def resize_half(img):
    img = _f32(img)
    r, c = C.c_int(), C.c_int()
    lib().orc_resize_half(_p(img), img.shape[0], img.shape[1], None, C.byref(r), C.byref(c))
    out = np.empty((r.value, c.value), np.float32)
    lib().orc_resize_half(_p(img), img.shape[0], img.shape[1], _p(out), C.byref(r), C.byref(c))
    return out


def hessian_response(img, norm):
    img = _f32(img)
    out = np.empty_like(img)
    lib().orc_hessian_response(_p(img), img.shape[0], img.shape[1], C.c_float(norm), _p(out))
    return out


def interpolate(img, ofsx, ofsy, a11, a12, a21, a22, rrows, rcols):
    img = _f32(img)
    out = np.empty((rrows, rcols), np.float32)
    t = lib().orc_interpolate(_p(img), img.shape[0], img.shape[1], C.c_float(ofsx), C.c_float(ofsy), C.c_float(a11),
                              C.c_float(a12), C.c_float(a21), C.c_float(a22), _p(out), rrows, rcols)
    return out, bool(t)


def atan_lut():
    return np.ctypeslib.as_array(lib().orc_atan_lut(), shape=(256,)).copy()


def atan2lut(y, x):
    return lib().orc_atan2lut(C.c_float(y), C.c_float(x))


def octave_levels(first, params, nlevels=5):
    first = _f32(first)
    blurs = np.empty((nlevels,) + first.shape, np.float32)
    resps = np.empty((nlevels,) + first.shape, np.float32)
    lib().orc_octave_levels(_p(first), first.shape[0], first.shape[1], C.byref(params), _p(blurs), _p(resps))
    return blurs, resps


def detect_scalespace(img, params, cap=400000):
    img = _f32(img)
    out = np.zeros(cap, SSKP)
    n = lib().orc_detect_scalespace(_p(img), img.shape[0], img.shape[1], C.byref(params), _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def response(img, detector_type, norm):
    """ScaleSpaceDetector::Response (pyramid.cpp:132-175): 0 Hessian, 1 DoG, 2 Harris."""
    img = _f32(img)
    out = np.zeros_like(img)
    lib().orc_response(_p(img), img.shape[0], img.shape[1], int(detector_type), C.c_float(norm), _p(out))
    return out


def detect_hessaff(img, params, tilt=1.0, zoom=1.0, cap=400000):
    img = _f32(img)
    out = np.zeros(cap, KEYPOINT)
    n = lib().orc_detect_hessaff(_p(img), img.shape[0], img.shape[1], C.byref(params), C.c_double(tilt),
                                 C.c_double(zoom), _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def find_affine_shape(blur, params, x, y, s, pixel_distance):
    blur = _f32(blur)
    u = np.zeros(4, np.float32)
    ok = lib().orc_find_affine_shape(_p(blur), blur.shape[0], blur.shape[1], C.byref(params), C.c_float(x),
                                     C.c_float(y), C.c_float(s), C.c_float(pixel_distance), _p(u))
    return ok, u


def detect_affine_regions(kps, img_id=0, det_type=0):
    kps = np.ascontiguousarray(kps, KEYPOINT)
    out = np.zeros(len(kps), REGION)
    lib().orc_detect_affine_regions(_p(kps), len(kps), img_id, det_type, _p(out))
    return out


def detect_orientation(img, regs, mr_size=1.0, patch_size=41, half=0, max_ang=1, th=0.8, upright=0):
    img = _f32(img)
    regs = np.ascontiguousarray(regs, REGION)
    cap = max(1, len(regs) * max(1, 36 if max_ang < 0 else max_ang) + len(regs))
    out = np.zeros(cap, REGION)
    n = lib().orc_detect_orientation(_p(img), img.shape[0], img.shape[1], _p(regs), len(regs), C.c_double(mr_size),
                                     patch_size, half, max_ang, C.c_double(th), upright, _p(out), cap)
    assert n <= cap
    return out[:n].copy()


def dominant_angles(patch, half=0, th=0.8, max_angles=-1):
    patch = _f32(patch)
    out = np.zeros(40, np.float32)
    n = lib().orc_dominant_angles(_p(patch), patch.shape[0], half, C.c_double(th), max_angles, _p(out), 40)
    return out[:n].copy()


def reproject_regions(regs, H, w, h):
    regs = np.ascontiguousarray(regs, REGION).copy()
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    n = lib().orc_reproject_regions(_p(regs), len(regs), _p(H), w, h)
    return regs[:n].copy()


def reproject_regions_touch_boundary(regs, H, w, h, mr_size=3.0 * 3.0 ** 0.5):
    regs = np.ascontiguousarray(regs, REGION).copy()
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    n = lib().orc_reproject_regions_touch_boundary(_p(regs), len(regs), _p(H), w, h, C.c_double(mr_size))
    return regs[:n].copy()


def describe_regions(img, regs, mr_size=5.1962, patch_size=41, fast=0, photo_norm=1, rootsift=1, max_bin=0.2):
    img = _f32(img)
    regs = np.ascontiguousarray(regs, REGION)
    desc = np.zeros((len(regs), 128), np.float32)
    lib().orc_describe_regions(_p(img), img.shape[0], img.shape[1], _p(regs), len(regs), C.c_double(mr_size),
                               patch_size, fast, photo_norm, rootsift, C.c_double(max_bin), _p(desc))
    return desc


def extract_patch(img, reg, mr_size=5.1962, patch_size=41):
    img = _f32(img)
    reg = np.ascontiguousarray(reg, REGION).reshape(1)
    patch = np.zeros((patch_size, patch_size), np.float32)
    lib().orc_extract_patch(_p(img), img.shape[0], img.shape[1], _p(reg), C.c_double(mr_size), patch_size, _p(patch))
    return patch


def describe_patch(patch, photo_norm=1, rootsift=1, max_bin=0.2):
    p = _f32(patch).copy()
    desc = np.zeros(128, np.float32)
    lib().orc_describe_patch(_p(p), photo_norm, rootsift, C.c_double(max_bin), _p(desc))
    return desc, p


def knn_linear(d1, d2, nn=50):
    d1, d2 = _f32(d1), _f32(d2)
    idx = np.zeros((len(d1), nn), np.int32)
    dist = np.zeros((len(d1), nn), np.float32)
    lib().orc_knn_linear(_p(d1), len(d1), _p(d2), len(d2), d1.shape[1], nn, _p(idx), _p(dist))
    return idx, dist


def match_fginn(d1, d2, pos2, ratio=0.8, contrad_dist=30.0, nn=50):
    d1, d2 = _f32(d1), _f32(d2)
    pos2 = np.ascontiguousarray(pos2, np.float64)
    out = np.zeros(max(1, len(d1)), TENT)
    n = lib().orc_match_fginn(_p(d1), len(d1), _p(d2), len(d2), d1.shape[1] if len(d1) else 128, _p(pos2),
                              C.c_double(ratio), C.c_double(contrad_dist), nn, _p(out), len(out))
    return out[:n].copy()


def duplicate_filtering(pts, key, r=2.0, do_sort=True):
    pts = np.ascontiguousarray(pts, np.float64)
    key = np.ascontiguousarray(key, np.float64)
    T = len(pts)
    order = np.zeros(T, np.int32)
    keep = np.zeros(T, np.uint8)
    lib().orc_duplicate_filtering(_p(pts), _p(key), T, C.c_double(r), int(do_sort), _p(order), _p(keep))
    return order, keep.astype(bool)


def ref_available():
    return bool(lib().orc_ref_available())


def detect_msers(img, min_size=30, max_area=0.05, min_margin=8.0, relative=0, mode=0, reg_number=500, rel_threshold=-1.0,
                 rel_reg_number=-1.0, tilt=1.0, zoom=1.0):
    """DetectMSERs (extrema.cpp:284-473, doOnNormal); defaults = [MSER] of config_iter_mods_cviu.ini.  PARITY UNPINNED."""
    img = _f32(img)
    cap = 1 << 16
    while True:
        out = np.zeros(cap, KEYPOINT)
        n = lib().orc_detect_msers(_p(img), img.shape[0], img.shape[1], int(min_size), C.c_double(max_area),
                                   C.c_double(min_margin), int(relative), int(mode), int(reg_number),
                                   C.c_double(rel_threshold), C.c_double(rel_reg_number), C.c_double(tilt),
                                   C.c_double(zoom), _p(out), cap)
        if n <= cap:
            return out[:n].copy()
        cap = n


def loransac_h(pts, laf1, laf2, err_threshold=3.0, confidence=0.99, max_samples=100000, lo=1, hlaf_coef=12.0,
               sym_check=1, seed=1, error_type=0):
    pts = np.ascontiguousarray(pts, np.float64)
    laf1 = np.ascontiguousarray(laf1, np.float64)
    laf2 = np.ascontiguousarray(laf2, np.float64)
    T = len(pts)
    H = np.zeros(9)
    Hraw = np.zeros(9)
    inl = np.zeros(max(T, 1), np.uint8)
    keep = np.zeros(max(T, 1), np.uint8)
    dout = np.zeros(3, np.int32)
    n = lib().orc_loransac_h(_p(pts), _p(laf1), _p(laf2), T, C.c_double(err_threshold), C.c_double(confidence),
                             max_samples, lo, C.c_double(hlaf_coef), sym_check, C.c_uint(seed), _p(H), _p(Hraw),
                             _p(inl), _p(keep), _p(dout), int(error_type))
    return dict(n=n, H=H.reshape(3, 3), Hraw=Hraw, inl=inl[:T].astype(bool), keep=keep[:T].astype(bool),
                samples=int(dout[0]), lo_count=int(dout[1]), ori_rejects=int(dout[2]))


def loransac_f(pts, laf1, laf2, err_threshold=4.0, confidence=0.99, max_samples=100000, lo=1, laf_coef=3.0,
               sym_check=1, error_type=0, seed=1):
    """LORANSACFiltering with useF = 1 (the reference's exp_ransacFcustom from oracle/_ref + F_LAF_check)."""
    pts = np.ascontiguousarray(pts, np.float64)
    laf1 = np.ascontiguousarray(laf1, np.float64)
    laf2 = np.ascontiguousarray(laf2, np.float64)
    T = len(pts)
    F = np.zeros(9)
    inl = np.zeros(max(T, 1), np.uint8)
    keep = np.zeros(max(T, 1), np.uint8)
    dout = np.zeros(3, np.int32)
    n = lib().orc_loransac_f(_p(pts), _p(laf1), _p(laf2), T, C.c_double(err_threshold), C.c_double(confidence),
                             max_samples, lo, C.c_double(laf_coef), sym_check, error_type, C.c_uint(seed), _p(F),
                             _p(inl), _p(keep), _p(dout))
    return dict(n=n, F=F.reshape(3, 3), inl=inl[:T].astype(bool), keep=keep[:T].astype(bool),
                samples=int(dout[0]), lo_count=int(dout[1]))


def warp_affine(img, M6, drows, dcols, border=128.0):
    img = _f32(img)
    M6 = np.ascontiguousarray(M6, np.float64).reshape(6)
    out = np.empty((drows, dcols), np.float32)
    lib().orc_warp_affine(_p(img), img.shape[0], img.shape[1], _p(M6), _p(out), drows, dcols, C.c_float(border))
    return out


def gaussian_blur_xy(img, kx, ky, sx, sy):
    img = _f32(img)
    out = np.empty_like(img)
    lib().orc_gaussian_blur_xy(_p(img), img.shape[0], img.shape[1], kx, ky, C.c_double(sx), C.c_double(sy), _p(out))
    return out


def make_view(tilt=1.0, phi=0.0, zoom=1.0, init_sigma=0.5, do_blur=1):
    v = View()
    v.zoom, v.tilt, v.phi, v.InitSigma, v.doBlur = zoom, tilt, phi, init_sigma, do_blur
    return v


def synth_view(gray, view):
    gray = _f32(gray)
    r, c = C.c_int(), C.c_int()
    H = np.zeros(9)
    lib().orc_synth_view(_p(gray), gray.shape[0], gray.shape[1], C.byref(view), None, C.byref(r), C.byref(c), _p(H))
    out = np.empty((r.value, c.value), np.float32)
    ident = lib().orc_synth_view(_p(gray), gray.shape[0], gray.shape[1], C.byref(view), _p(out), C.byref(r),
                                 C.byref(c), _p(H))
    return out, H.reshape(3, 3), bool(ident)


def set_vs_pars(scale_set, tilt_set, phi_base, init_sigma=0.5, do_blur=1, prev=None):
    """SetVSPars; prev is a python list of View that is extended in place (de-duplication across steps)."""
    prev = [] if prev is None else prev
    ss = np.ascontiguousarray(scale_set, np.float64)
    ts = np.ascontiguousarray(tilt_set, np.float64)
    cap = 4096
    par = (View * cap)()
    pv = (View * cap)(*prev)
    npv = C.c_int(len(prev))
    n = lib().orc_set_vs_pars(_p(ss), len(ss), _p(ts), len(ts), C.c_double(phi_base), C.c_double(init_sigma),
                              int(do_blur), par, cap, pv, C.byref(npv), cap)
    del prev[:]
    for i in range(npv.value):
        prev.append(make_view(pv[i].tilt, pv[i].phi, pv[i].zoom, pv[i].InitSigma, pv[i].doBlur))
    return [make_view(par[i].tilt, par[i].phi, par[i].zoom, par[i].InitSigma, par[i].doBlur) for i in range(n)]


def detect_describe_views(gray, views, params=None, ori=(1.0, 41, 1, 0.8), desc=(5.1962, 41, 0, 1, 1, 0.2), mser=None, threads=1,
                          descs=None):
    """The HessianAffine / MSER branch of SynthDetectDescribeKeypoints (imagerepresentation.cpp:603-2047): per view
    synthesise, detect, orient, reproject, describe; concatenate in view order with AddRegionsToList id re-basing
    (:588-600).  Returns (regions, descriptors).  threads > 1: the views run on a thread pool (the reference's `#pragma omp
    parallel for` over views, :612-622; ctypes releases the GIL), same result.
    descs = the step's descriptor list (types 0 SIFT, 1 RootSIFT, 2 HalfSIFT, 3 HalfRootSIFT; default [desc[4]]): the
    reference orients ONCE per step -- with doHalfSIFT = true as soon as one name contains "Half" (:693-706, 1259-1264) -- and
    describes every descriptor of the step on that oriented list (:1288-1296).  With `descs` the second return value is the
    list of descriptor arrays, one per entry."""
    gray = _f32(gray)
    params = params or default_params()
    types = [desc[4]] if descs is None else list(descs)
    half = 1 if any(t >= 2 for t in types) else 0

    def one(job):
        vi, v = job
        img, H, ident = synth_view(gray, v)
        vt, vz = (abs(v.tilt) if not ident else 1.0), (v.zoom if not ident else 1.0)
        if mser is not None:   # DetectAffineRegions(..., DET_MSER, DetectMSERs), imagerepresentation.cpp:1037
            k = detect_msers(img, tilt=vt, zoom=vz, **mser)
            regs = detect_affine_regions(k, img_id=0 if ident else vi, det_type=3)
        else:
            k = detect_hessaff(img, params, tilt=vt, zoom=vz)
            regs = detect_affine_regions(k, img_id=0 if ident else vi)
        ro = detect_orientation(img, regs, mr_size=ori[0], patch_size=ori[1], half=half, max_ang=ori[2], th=ori[3])
        rr = reproject_regions(ro, H.reshape(9), gray.shape[1], gray.shape[0])
        d = [describe_regions(img, rr, mr_size=desc[0], patch_size=desc[1], fast=desc[2], photo_norm=desc[3],
                              rootsift=t, max_bin=desc[5]) for t in types]
        return rr.copy(), d

    jobs = list(enumerate(views))
    if threads > 1 and len(jobs) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(threads) as pool:
            parts = list(pool.map(one, jobs))
    else:
        parts = [one(j) for j in jobs]
    all_regs, all_desc = [], []
    size = 0
    for rr, d in parts:
        rr["id"] += size
        rr["parent_id"] += size
        size += len(rr)
        all_regs.append(rr)
        all_desc.append(d)
    regs = np.concatenate(all_regs) if all_regs else np.zeros(0, REGION)
    per = [np.concatenate([d[k] for d in all_desc]) if all_desc else np.zeros((0, 128), np.float32) for k in range(len(types))]
    return regs, (per if descs is not None else per[0])

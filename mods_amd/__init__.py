"""mods_amd -- Python harness over the C ABI of libmodsx.so (include/modsx.h).

The product is the shared library (HIP kernels + C++ host engine); this module only
binds it with ctypes so the tests and bench.py can drive it.  There is no Python or
CPU implementation of the path in here: without the built library, or without a GPU,
every device entry point raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MODSX_LIB", os.path.join(_HERE, "libmodsx.so"))   # MODSX_LIB: kernel experiments (tools/)

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


class HessAffParams(C.Structure):
    _fields_ = [("threshold", C.c_float), ("mode", C.c_int), ("reg_number", C.c_int),
                ("rel_threshold", C.c_float), ("rel_reg_number", C.c_float), ("numberOfScales", C.c_int),
                ("initialSigma", C.c_float), ("edgeEigenValueRatio", C.c_double), ("border", C.c_int),
                ("maxIterations", C.c_int), ("convergenceThreshold", C.c_float), ("smmWindowSize", C.c_int),
                ("affInitialSigma", C.c_float), ("doBaumberg", C.c_int), ("detectorType", C.c_int)]


class MserParams(C.Structure):
    _fields_ = [("min_size", C.c_int), ("max_area", C.c_double), ("min_margin", C.c_double), ("relative", C.c_int),
                ("mode", C.c_int), ("reg_number", C.c_int), ("rel_threshold", C.c_float), ("rel_reg_number", C.c_float)]


class PairParams(C.Structure):
    _fields_ = [("det", HessAffParams),
                ("ori_mrSize", C.c_double), ("ori_patchSize", C.c_int), ("ori_maxAngles", C.c_int),
                ("ori_threshold", C.c_double),
                ("desc_mrSize", C.c_double), ("desc_patchSize", C.c_int), ("desc_photoNorm", C.c_int),
                ("desc_type", C.c_int), ("desc_maxBinValue", C.c_double),
                ("match_ratio", C.c_double), ("contradDist", C.c_double), ("nn", C.c_int),
                ("duplicateDist", C.c_double),
                ("err_threshold", C.c_double), ("confidence", C.c_double), ("max_samples", C.c_int),
                ("localOptimization", C.c_int), ("HLAFCoef", C.c_double), ("doSymmCheck", C.c_int),
                ("ransac_seed", C.c_uint), ("useF", C.c_int), ("LAFCoef", C.c_double), ("errorType", C.c_int),
                ("detector", C.c_int), ("mser", MserParams),
                ("n_desc", C.c_int), ("desc_types", C.c_int * 4), ("desc_ratios", C.c_double * 4)]


class View(C.Structure):
    _fields_ = [("zoom", C.c_double), ("tilt", C.c_double), ("phi", C.c_double), ("InitSigma", C.c_double),
                ("doBlur", C.c_int)]


def make_view(tilt=1.0, phi=0.0, zoom=1.0, init_sigma=0.5, do_blur=1):
    v = View()
    v.zoom, v.tilt, v.phi, v.InitSigma, v.doBlur = zoom, tilt, phi, init_sigma, do_blur
    return v


def _view_array(views):
    arr = (View * len(views))()
    for i, v in enumerate(views):
        arr[i].zoom, arr[i].tilt, arr[i].phi, arr[i].InitSigma, arr[i].doBlur = v.zoom, v.tilt, v.phi, v.InitSigma, v.doBlur
    return arr


class LadderStep(C.Structure):
    _fields_ = [("views", C.c_void_p), ("nviews", C.c_int), ("match_ratio", C.c_double), ("detector", C.c_int),
                ("n_desc", C.c_int), ("desc_types", C.c_int * 4), ("desc_ratios", C.c_double * 4)]


class PairResult(C.Structure):
    _fields_ = [("n_regions1", C.c_int), ("n_regions2", C.c_int), ("n_tentatives", C.c_int), ("n_unique", C.c_int),
                ("n_ransac_inliers", C.c_int), ("n_verified", C.c_int), ("ransac_samples", C.c_int),
                ("ransac_lo", C.c_int), ("H", C.c_double * 9), ("tentatives", C.c_void_p),
                ("ransac_inlier", C.c_void_p), ("verified", C.c_void_p)]


EXPORTS = ["modsx_version", "modsx_last_error", "modsx_free", "modsx_create", "modsx_destroy", "modsx_synchronize",
           "modsx_default_hessaff_params", "modsx_default_pair_params", "modsx_image_upload", "modsx_image_update",
           "modsx_image_wrap_device", "modsx_image_free", "modsx_image_download", "modsx_detect_affine_keypoints",
           "modsx_detect_scalespace", "modsx_octave_levels", "modsx_gaussian_blur", "modsx_resize_half", "modsx_response",
           "modsx_detect_affine_regions", "modsx_detect_orientation", "modsx_reproject_regions", "modsx_reproject_regions_touch_boundary",
           "modsx_describe_regions", "modsx_match_fginn", "modsx_duplicate_filtering", "modsx_ransac_h",
           "modsx_loransac_h", "modsx_ransac_h_errtype", "modsx_loransac_h_errtype", "modsx_ransac_f", "modsx_loransac_f", "modsx_match_pair", "modsx_match_pairs", "modsx_match_pairs_views", "modsx_pair_result_release",
           "modsx_set_vs_pars", "modsx_synth_view", "modsx_detect_describe_views", "modsx_match_fginn_device",
           "modsx_match_pair_views", "modsx_match_ladder", "modsx_save_regions", "modsx_load_regions", "modsx_default_mser_params", "modsx_detect_msers", "modsx_detect_msers_u8", "modsx_last_timings", "modsx_profile",
           "modsx_kernel_stats", "modsx_last_batch_verify", "modsx_comm_unique_id", "modsx_comm_create", "modsx_comm_destroy", "modsx_comm_info",
           "modsx_view_block_order", "modsx_detect_describe_views_sharded", "modsx_match_fginn_sharded",
           "modsx_match_pair_views_sharded", "modsx_match_pairs_views_sharded", "modsx_match_ladder_sharded", "modsx_comm_loopback_id", "modsx_comm_set_lanes",
           "modsx_comm_attach", "modsx_comm_lane_done", "modsx_comm_reset_lanes", "modsx_comm_set_timeout", "modsx_comm_stats",
           "modsx_shard_block_bytes", "modsx_shard_block_pack", "modsx_shard_blocks_unpack", "modsx_shard_device_pack",
           "modsx_shard_device_unpack", "modsx_verify_device_stats", "modsx_verify_device_timing", "modsx_comm_set_exchange",
           "modsx_shard_owner_plan"]
# include/modsx_degensac.h: the reference's own verification symbols (link-time drop-in for libdegensac)
EXPORTS_DEGENSAC = ["exp_ransacHcustom", "exp_ransacFcustom", "HDs", "HDsi", "HDsidx", "HDsSym", "HDsiSym", "HDsSymidx",
                    "HDsSymMax", "HDsiSymMax", "HDsSymidxMax", "FDs", "FDsSym", "exFDs", "exFDsSym",
                    "modsx_ransac_set_seed"]

SHARD_ROW_REGION, SHARD_ROW_KP = 0, 1     # include/modsx.h: what of a region travels in a row (all 200 B / the 56 B verification slice)
EXCHANGE_ALL_GATHER, EXCHANGE_OWNER = 0, 1   # include/modsx.h: modsx_comm_set_exchange
KP_FIELDS = ("x", "y", "a11", "a12", "a21", "a22", "s")
KERNEL_CLASSES = ["blur_hess", "hessian", "resize", "nms_localize", "baumberg", "orientation", "patch_sample",
                  "blur_rows", "describe", "match_fginn", "gray", "warp_affine", "view_blur", "blur_cols", "match_sweep1"]


def build(force=False):
    """Compile libmodsx.so for gfx950 with hipcc (in-tree)."""
    src = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src, f) for f in os.listdir(src) if f.endswith((".hip", ".cpp", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "modsx.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", src, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libmodsx.so is not built; run __graft_entry__.build() (needs hipcc)")
        L = C.CDLL(LIB_PATH)
        L.modsx_last_error.restype = C.c_char_p
        L.modsx_create.restype = C.c_void_p
        L.modsx_create.argtypes = [C.c_int]
        L.modsx_comm_create.restype = C.c_void_p
        L.modsx_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.modsx_comm_destroy.argtypes = [C.c_void_p]
        L.modsx_comm_info.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.modsx_comm_loopback_id.argtypes = [C.c_void_p, C.c_int]
        L.modsx_comm_set_lanes.argtypes = [C.c_void_p, C.c_int]
        L.modsx_comm_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.modsx_comm_lane_done.argtypes = [C.c_void_p, C.c_int]
        L.modsx_comm_reset_lanes.argtypes = [C.c_void_p]
        L.modsx_comm_set_timeout.argtypes = [C.c_void_p, C.c_int]
        L.modsx_comm_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.modsx_destroy.argtypes = [C.c_void_p]
        L.modsx_free.argtypes = [C.c_void_p]
        L.modsx_image_upload.restype = C.c_void_p
        L.modsx_image_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.modsx_image_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.modsx_debug_last_batch_verify_each.argtypes = [C.c_void_p, C.c_int]
        L.modsx_image_wrap_device.restype = C.c_void_p
        L.modsx_image_wrap_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.modsx_image_free.argtypes = [C.c_void_p, C.c_void_p]
        L.modsx_synth_view.restype = C.c_void_p
        L.modsx_synth_view.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _err():
    return lib().modsx_last_error().decode()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _check(rc, what):
    if rc < 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, _err()))
    return rc


def default_hessaff_params(**kw):
    p = HessAffParams()
    lib().modsx_default_hessaff_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_mser_params(**kw):
    p = MserParams()
    lib().modsx_default_mser_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def detect_msers_u8(gray, params=None, tilt=1.0, zoom=1.0):
    """DetectMSERs on a host u8 image (no device needed)."""
    params = params or default_mser_params()
    g = np.ascontiguousarray(gray, np.uint8)
    out = C.c_void_p()
    n = _check(lib().modsx_detect_msers_u8(_p(g), g.shape[0], g.shape[1], C.byref(params), C.c_double(tilt),
                                           C.c_double(zoom), C.byref(out)), "detect_msers_u8")
    return _take(out, n, KEYPOINT)


def default_pair_params(**kw):
    p = PairParams()
    lib().modsx_default_pair_params(C.byref(p))
    descs = kw.pop("descs", None)     # [(MODSX_DESC_* type, FGINN ratio), ...]: the step's Descriptors / FGINNThreshold lists
    if descs is not None:
        _set_descs(p, descs)
    for k, v in kw.items():
        if hasattr(p.det, k) and not hasattr(p, k):
            setattr(p.det, k, v)
        else:
            setattr(p, k, v)
    return p


def _set_descs(obj, descs):
    obj.n_desc = len(descs)
    for i, (t, r) in enumerate(descs):
        obj.desc_types[i] = int(t)
        obj.desc_ratios[i] = float(r)


def _take(ptr, n, dtype):
    """Copy a malloc'd array returned through `**out` into numpy and free it."""
    if n <= 0:
        if ptr:
            lib().modsx_free(ptr)
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr.value if hasattr(ptr, "value") else ptr)
    arr = np.frombuffer(buf, dtype=dtype, count=n).copy()
    lib().modsx_free(ptr)
    return arr


# ---- host-only entry points (no device needed) ---------------------------------------------------
def detect_affine_regions(kps, img_id=0, det_type=0):
    kps = np.ascontiguousarray(kps, KEYPOINT)
    out = np.zeros(len(kps), REGION)
    _check(lib().modsx_detect_affine_regions(_p(kps), len(kps), img_id, det_type, _p(out)), "detect_affine_regions")
    return out


def reproject_regions(regs, H, w, h):
    regs = np.ascontiguousarray(regs, REGION).copy()
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    n = _check(lib().modsx_reproject_regions(_p(regs), len(regs), _p(H), int(w), int(h)), "reproject_regions")
    return regs[:n].copy()


def reproject_regions_touch_boundary(regs, H, w, h, mr_size=3.0 * 3.0 ** 0.5):
    regs = np.ascontiguousarray(regs, REGION).copy()
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    n = _check(lib().modsx_reproject_regions_touch_boundary(_p(regs), len(regs), _p(H), int(w), int(h), C.c_double(mr_size)),
               "reproject_regions_touch_boundary")
    return regs[:n].copy()


def duplicate_filtering(pts, key, r=2.0, do_sort=True):
    pts = np.ascontiguousarray(pts, np.float64)
    key = np.ascontiguousarray(key, np.float64)
    T = len(pts)
    order = np.zeros(max(T, 1), np.int32)
    keep = np.zeros(max(T, 1), np.uint8)
    _check(lib().modsx_duplicate_filtering(_p(pts), _p(key), T, C.c_double(r), int(do_sort), _p(order), _p(keep)),
           "duplicate_filtering")
    return order[:T], keep[:T].astype(bool)


def ransac_h(u, th, conf=0.99, max_sam=100000, oriented=1, sym_check=1, seed=1):
    u = np.ascontiguousarray(u, np.float64)
    n = len(u)
    H = np.zeros(9)
    inl = np.zeros(n, np.uint8)
    dout = np.zeros(3, np.int32)
    J = C.c_double(0)
    rc = _check(lib().modsx_ransac_h(_p(u), n, C.c_double(th), C.c_double(conf), int(max_sam), _p(H), _p(inl), _p(dout),
                                     int(oriented), int(sym_check), C.c_uint(seed), C.byref(J)), "ransac_h")
    return dict(n=rc, H=H, inl=inl.astype(bool), samples=int(dout[0]), lo_count=int(dout[1]),
                ori_rejects=int(dout[2]), J=J.value)


def loransac_h(pts, laf1, laf2, err_threshold=3.0, confidence=0.99, max_samples=100000, lo=1, hlaf_coef=12.0,
               sym_check=1, seed=1, error_type=0):
    pts = np.ascontiguousarray(pts, np.float64)
    laf1 = np.ascontiguousarray(laf1, np.float64)
    laf2 = np.ascontiguousarray(laf2, np.float64)
    T = len(pts)
    H, Hraw = np.zeros(9), np.zeros(9)
    inl = np.zeros(max(T, 1), np.uint8)
    keep = np.zeros(max(T, 1), np.uint8)
    dout = np.zeros(3, np.int32)
    n = _check(lib().modsx_loransac_h_errtype(_p(pts), _p(laf1), _p(laf2), T, C.c_double(err_threshold),
                                              C.c_double(confidence), int(max_samples), int(lo), C.c_double(hlaf_coef),
                                              int(sym_check), int(error_type), C.c_uint(seed), _p(H), _p(Hraw), _p(inl),
                                              _p(keep), _p(dout)), "loransac_h")
    return dict(n=n, H=H.reshape(3, 3), Hraw=Hraw, inl=inl[:T].astype(bool), keep=keep[:T].astype(bool),
                samples=int(dout[0]), lo_count=int(dout[1]), ori_rejects=int(dout[2]))


def loransac_f(pts, laf1, laf2, err_threshold=4.0, confidence=0.99, max_samples=100000, lo=1, laf_coef=3.0,
               sym_check=1, error_type=0, seed=1):
    """LORANSACFiltering with useF = 1 (exp_ransacFcustom + F_LAF_check), host C++."""
    pts = np.ascontiguousarray(pts, np.float64)
    laf1 = np.ascontiguousarray(laf1, np.float64)
    laf2 = np.ascontiguousarray(laf2, np.float64)
    T = len(pts)
    F = np.zeros(9)
    inl = np.zeros(max(T, 1), np.uint8)
    keep = np.zeros(max(T, 1), np.uint8)
    dout = np.zeros(3, np.int32)
    n = _check(lib().modsx_loransac_f(_p(pts), _p(laf1), _p(laf2), T, C.c_double(err_threshold),
                                      C.c_double(confidence), int(max_samples), int(lo), C.c_double(laf_coef),
                                      int(sym_check), int(error_type), C.c_uint(seed), _p(F), _p(inl), _p(keep),
                                      _p(dout)), "loransac_f")
    return dict(n=n, F=F.reshape(3, 3), inl=inl[:T].astype(bool), keep=keep[:T].astype(bool),
                samples=int(dout[0]), lo_count=int(dout[1]), degen_count=int(dout[2]))


def verify_device_stats(reset=False):
    """modsx_verify_device_stats: how much of DEGENSAC's rFtH hypothesis loop ran on the device (process-wide counters)."""
    out = (C.c_long * 6)()
    lib().modsx_verify_device_stats(out, int(bool(reset)))
    d = dict(zip(("batches", "hypotheses", "events", "disagreements", "loops", "loop_us"), list(out)))
    t = (C.c_long * 4)()
    lib().modsx_verify_device_timing(t, int(bool(reset)))
    d.update(zip(("draw_us", "device_wait_us", "host_phase_us", "event_body_us"), list(t)))
    return d


class RegionClass(C.Structure):
    _fields_ = [("det_name", C.c_char_p), ("desc_name", C.c_char_p), ("regs", C.c_void_p), ("desc", C.c_void_p),
                ("n", C.c_int), ("dim", C.c_int), ("stride", C.c_int)]


def save_regions(path, classes):
    """ImageRepresentation::SaveRegions: classes = [(det_name, desc_name, regs (REGION), desc [n, stride] f32, dim)]."""
    arr = (RegionClass * len(classes))()
    keep = []
    for i, (det, dn, regs, desc, dim) in enumerate(classes):
        regs = np.ascontiguousarray(regs, REGION)
        desc = np.ascontiguousarray(desc, np.float32).reshape(len(regs), -1) if len(regs) else np.zeros((0, max(dim, 1)), np.float32)
        keep += [regs, desc]
        arr[i].det_name, arr[i].desc_name = det.encode(), dn.encode()
        arr[i].regs, arr[i].desc = regs.ctypes.data, desc.ctypes.data
        arr[i].n, arr[i].dim, arr[i].stride = len(regs), dim, desc.shape[1]
    _check(lib().modsx_save_regions(path.encode(), arr, len(classes)), "save_regions")


def load_regions(path, det_name="", desc_name=""):
    """ImageRepresentation::LoadRegions for one class: returns (det_name, desc_name, regs, desc [n, dim])."""
    regs, desc = C.c_void_p(), C.c_void_p()
    dim = C.c_int(0)
    fd, fs = C.create_string_buffer(64), C.create_string_buffer(64)
    n = _check(lib().modsx_load_regions(path.encode(), det_name.encode(), desc_name.encode(), C.byref(regs), C.byref(desc),
                                        C.byref(dim), fd, fs), "load_regions")
    r = _take(regs, n, REGION)
    d = _take(desc, n * dim.value, np.dtype(np.float32)).reshape(n, dim.value)
    return fd.value.decode(), fs.value.decode(), r, d


def set_vs_pars(scale_set, tilt_set, phi_base, init_sigma=0.5, do_blur=1, prev=None):
    """SetVSPars (host): returns the new views of this step; `prev` (list of View) is extended in place."""
    prev = [] if prev is None else prev
    ss = np.ascontiguousarray(scale_set, np.float64)
    ts = np.ascontiguousarray(tilt_set, np.float64)
    cap = 4096
    par = (View * cap)()
    pv = (View * cap)(*prev)
    npv = C.c_int(len(prev))
    n = _check(lib().modsx_set_vs_pars(_p(ss), len(ss), _p(ts), len(ts), C.c_double(phi_base), C.c_double(init_sigma),
                                       int(do_blur), par, cap, pv, C.byref(npv), cap), "set_vs_pars")
    del prev[:]
    for i in range(npv.value):
        prev.append(make_view(pv[i].tilt, pv[i].phi, pv[i].zoom, pv[i].InitSigma, pv[i].doBlur))
    return [make_view(par[i].tilt, par[i].phi, par[i].zoom, par[i].InitSigma, par[i].doBlur) for i in range(n)]


# ---- device path -------------------------------------------------------------------------------
class Image(object):
    def __init__(self, ctx, handle, rows, cols):
        self.ctx, self.h, self.rows, self.cols = ctx, handle, rows, cols

    def free(self):
        if self.h:
            lib().modsx_image_free(self.ctx.h, self.h)
            self.h = None

    def update(self, pixels):
        """modsx_image_update: new pixels (u8 or f32, 1 or 3 channels) for this uploaded image, same size, no allocation."""
        a = np.ascontiguousarray(pixels)
        if a.dtype != np.uint8:
            a = np.ascontiguousarray(a, np.float32)
        ch = 1 if a.ndim == 2 else a.shape[2]
        _check(lib().modsx_image_update(C.c_void_p(self.ctx.h), C.c_void_p(self.h), _p(a), a.shape[0], a.shape[1], ch,
                                        0 if a.dtype == np.uint8 else 1), "image_update")

    def download(self):
        out = np.empty((self.rows, self.cols), np.float32)
        _check(lib().modsx_image_download(C.c_void_p(self.ctx.h), C.c_void_p(self.h), _p(out)), "image_download")
        return out


class Context(object):
    """One modsx_ctx (one HIP stream).  Raises if no gfx950 device is available."""

    def __init__(self, device=0):
        self.h = lib().modsx_create(int(device))
        if not self.h:
            raise RuntimeError("modsx_create failed: " + _err())

    def close(self):
        if self.h:
            lib().modsx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _c(self):
        return C.c_void_p(self.h)

    def upload(self, pixels):
        a = np.ascontiguousarray(pixels)
        if a.dtype == np.uint8:
            dtype = 0
        else:
            a = np.ascontiguousarray(a, np.float32)
            dtype = 1
        ch = 1 if a.ndim == 2 else a.shape[2]
        h = lib().modsx_image_upload(self._c(), _p(a), a.shape[0], a.shape[1], ch, dtype)
        if not h:
            raise RuntimeError("modsx_image_upload failed: " + _err())
        return Image(self, h, a.shape[0], a.shape[1])

    def wrap_device(self, dev_ptr, rows, cols):
        h = lib().modsx_image_wrap_device(self._c(), C.c_void_p(dev_ptr), rows, cols)
        if not h:
            raise RuntimeError("modsx_image_wrap_device failed: " + _err())
        return Image(self, h, rows, cols)

    def synchronize(self):
        _check(lib().modsx_synchronize(self._c()), "synchronize")

    def gaussian_blur(self, img, sigma):
        out = np.empty((img.rows, img.cols), np.float32)
        _check(lib().modsx_gaussian_blur(self._c(), C.c_void_p(img.h), C.c_float(sigma), _p(out)), "gaussian_blur")
        return out

    def response(self, img, detector_type, norm):
        out = np.zeros((img.rows, img.cols), np.float32)
        _check(lib().modsx_response(self._c(), C.c_void_p(img.h), int(detector_type), C.c_float(norm), _p(out)), "response")
        return out

    def resize_half(self, img):
        r, c = C.c_int(), C.c_int()
        _check(lib().modsx_resize_half(self._c(), C.c_void_p(img.h), None, C.byref(r), C.byref(c)), "resize_half")
        out = np.empty((r.value, c.value), np.float32)
        _check(lib().modsx_resize_half(self._c(), C.c_void_p(img.h), _p(out), C.byref(r), C.byref(c)), "resize_half")
        return out

    def octave_levels(self, img, params):
        L = params.numberOfScales + 2
        blurs = np.empty((L, img.rows, img.cols), np.float32)
        resps = np.empty((L, img.rows, img.cols), np.float32)
        _check(lib().modsx_octave_levels(self._c(), C.c_void_p(img.h), C.byref(params), _p(blurs), _p(resps)),
               "octave_levels")
        return blurs, resps

    def detect_scalespace(self, img, params):
        out = C.c_void_p()
        n = _check(lib().modsx_detect_scalespace(self._c(), C.c_void_p(img.h), C.byref(params), C.byref(out)),
                   "detect_scalespace")
        return _take(out, n, SSKP)

    def detect_affine_keypoints(self, img, params, tilt=1.0, zoom=1.0):
        out = C.c_void_p()
        n = _check(lib().modsx_detect_affine_keypoints(self._c(), C.c_void_p(img.h), C.byref(params),
                                                       C.c_double(tilt), C.c_double(zoom), C.byref(out)),
                   "detect_affine_keypoints")
        return _take(out, n, KEYPOINT)

    def detect_orientation(self, img, regs, mr_size=1.0, patch_size=41, half=0, max_ang=1, th=0.8, upright=0):
        regs = np.ascontiguousarray(regs, REGION)
        out = C.c_void_p()
        n = _check(lib().modsx_detect_orientation(self._c(), C.c_void_p(img.h), _p(regs), len(regs),
                                                  C.c_double(mr_size), patch_size, half, max_ang, C.c_double(th),
                                                  upright, C.byref(out)), "detect_orientation")
        return _take(out, n, REGION)

    def describe_regions(self, img, regs, mr_size=5.1962, patch_size=41, fast=0, photo_norm=1, desc_type=1,
                         max_bin=0.2):
        regs = np.ascontiguousarray(regs, REGION)
        desc = np.zeros((len(regs), 128), np.float32)
        _check(lib().modsx_describe_regions(self._c(), C.c_void_p(img.h), _p(regs), len(regs), C.c_double(mr_size),
                                            patch_size, fast, photo_norm, desc_type, C.c_double(max_bin), _p(desc)),
               "describe_regions")
        return desc

    def match_fginn(self, d1, d2, pos2, ratio=0.8, contrad_dist=30.0, nn=50):
        d1 = np.ascontiguousarray(d1, np.float32)
        d2 = np.ascontiguousarray(d2, np.float32)
        pos2 = np.ascontiguousarray(pos2, np.float64)
        out = C.c_void_p()
        n = _check(lib().modsx_match_fginn(self._c(), _p(d1), len(d1), _p(d2), len(d2), _p(pos2), C.c_double(ratio),
                                           C.c_double(contrad_dist), nn, C.byref(out)), "match_fginn")
        return _take(out, n, TENT)

    def synth_view(self, img, view):
        H = np.zeros(9)
        ident = C.c_int(0)
        h = lib().modsx_synth_view(self._c(), C.c_void_p(img.h), C.byref(view), _p(H), C.byref(ident))
        if not h:
            raise RuntimeError("modsx_synth_view failed: " + _err())
        rows, cols = C.c_int(), C.c_int()
        out = Image(self, h, 0, 0)
        # query the size through the resize tap (cheap): rows/cols live in the handle; read them via download size
        out.rows, out.cols = _image_dims(h)
        return out, H.reshape(3, 3), bool(ident.value)

    def detect_describe_views(self, img, views, params, view_begin=0, view_step=1, want_desc=True, dev_desc=None,
                              dev_cap=0, want_counts=False):
        arr = _view_array(views)
        regs = C.c_void_p()
        desc = C.c_void_p()
        counts = (C.c_int * len(views))()
        n = _check(lib().modsx_detect_describe_views(self._c(), C.c_void_p(img.h), arr, len(views), C.byref(params),
                                                     int(view_begin), int(view_step), C.byref(regs),
                                                     C.byref(desc) if want_desc else None,
                                                     C.c_void_p(dev_desc) if dev_desc else None, C.c_long(dev_cap),
                                                     counts),
                   "detect_describe_views")
        r = _take(regs, n, REGION)
        d = None
        if want_desc:
            d = _take(desc, n * 128, np.dtype("f4")).reshape(n, 128) if n else np.zeros((0, 128), np.float32)
        if want_counts:
            return r, d, np.array(list(counts), np.int64)
        return r, d

    def match_fginn_device(self, d1_ptr, n1, d2_ptr, n2, pos2, ratio=0.8, contrad_dist=30.0, nn=50):
        pos2 = np.ascontiguousarray(pos2, np.float64)
        out = C.c_void_p()
        n = _check(lib().modsx_match_fginn_device(self._c(), C.c_void_p(d1_ptr), int(n1), C.c_void_p(d2_ptr), int(n2),
                                                  _p(pos2), C.c_double(ratio), C.c_double(contrad_dist), nn,
                                                  C.byref(out)), "match_fginn_device")
        return _take(out, n, TENT)

    def match_pair_views(self, img1, img2, views, params):
        arr = _view_array(views)
        res = PairResult()
        _check(lib().modsx_match_pair_views(self._c(), C.c_void_p(img1.h), C.c_void_p(img2.h), arr, len(views),
                                            C.byref(params), C.byref(res)), "match_pair_views")
        return _unpack_pair_result(res)

    # ---- view-sharded path (engine_shard.hip): `comm` is a handle from modsx_comm_create on this context ----
    def comm_create(self, id128, rank, world):
        h = lib().modsx_comm_create(self._c(), id128, int(rank), int(world))
        if not h:
            raise RuntimeError("modsx_comm_create failed: " + _err())
        return h

    def detect_describe_views_sharded(self, comm, img, views, params):
        """All regions in reference order on every rank + the device pointer of their [n][128] u8 descriptors."""
        arr = _view_array(views)
        regs, dptr = C.c_void_p(), C.c_void_p()
        counts = (C.c_int * len(views))()
        n = _check(lib().modsx_detect_describe_views_sharded(self._c(), C.c_void_p(comm), C.c_void_p(img.h), arr, len(views),
                                                             C.byref(params), C.byref(regs), C.byref(dptr), counts),
                   "detect_describe_views_sharded")
        return _take(regs, n, REGION), dptr.value, np.array(list(counts), np.int64)

    def match_fginn_sharded(self, comm, d1_ptr, n1, d2_ptr, n2, pos2, ratio=0.8, contrad_dist=30.0, nn=50):
        pos2 = np.ascontiguousarray(pos2, np.float64)
        out = C.c_void_p()
        n = _check(lib().modsx_match_fginn_sharded(self._c(), C.c_void_p(comm), C.c_void_p(d1_ptr), int(n1), C.c_void_p(d2_ptr),
                                                   int(n2), _p(pos2), C.c_double(ratio), C.c_double(contrad_dist), nn,
                                                   C.byref(out)), "match_fginn_sharded")
        return _take(out, n, TENT)

    def match_pair_views_sharded(self, comm, img1, img2, views, params, owner=0):
        arr = _view_array(views)
        res = PairResult()
        _check(lib().modsx_match_pair_views_sharded(self._c(), C.c_void_p(comm), C.c_void_p(img1.h), C.c_void_p(img2.h), arr,
                                                    len(views), C.byref(params), int(owner), C.byref(res)),
               "match_pair_views_sharded")
        return _unpack_pair_result(res)

    def match_pairs_views_sharded(self, comm, imgs1, imgs2, views, params, owner_base=0, arrays=True):
        """modsx_match_pairs_views_sharded: len(imgs1) <= 16 pairs in one sharded call; pair g is matched and verified by rank
        (owner_base + g) % world alone (owner_base < 0: every rank returns every pair).  Returns the list of per-pair results."""
        n = len(imgs1)
        a1 = (C.c_void_p * n)(*[im.h for im in imgs1])
        a2 = (C.c_void_p * n)(*[im.h for im in imgs2])
        arr = _view_array(views)
        res = (PairResult * n)()
        _check(lib().modsx_match_pairs_views_sharded(self._c(), C.c_void_p(comm), a1, a2, n, arr, len(views), C.byref(params),
                                                     int(owner_base), res), "match_pairs_views_sharded")
        return [_unpack_pair_result(res[i], arrays) for i in range(n)]

    def shard_device_pack(self, regs, descs, row_format=SHARD_ROW_REGION):
        """k_pack_rows on host-provided regions + descriptors: the rows part of a block, [n, R + 128 * ndesc] u8 (R = 200 or 56)."""
        L = lib()
        L.modsx_shard_device_pack.restype = C.c_long
        regs = np.ascontiguousarray(regs, REGION)
        descs = [np.ascontiguousarray(d, np.uint8) for d in descs]
        out = np.zeros((len(regs), shard_region_bytes(row_format) + 128 * len(descs)), np.uint8)
        _check(L.modsx_shard_device_pack(self._c(), _p(regs), _ptr_array(descs), len(descs), len(regs), int(row_format), _p(out)),
               "shard_device_pack")
        return out

    def shard_device_unpack(self, blocks, world, items, block_rows, ndesc=1, row_format=SHARD_ROW_REGION):
        """k_unpack_blocks on gathered blocks: (regs [world * block_rows] -- REGION records, or [.., 7] f64 for SHARD_ROW_KP --,
        [desc per class], pos [.., 2]); rows past the list's end are zero."""
        L = lib()
        L.modsx_shard_device_unpack.restype = C.c_long
        blocks = np.ascontiguousarray(blocks, np.uint8)
        cap = world * block_rows
        regs = np.zeros(cap, REGION) if row_format == SHARD_ROW_REGION else np.zeros((cap, 7), np.float64)
        descs = [np.zeros((cap, 128), np.uint8) for _ in range(ndesc)]
        pos = np.zeros((cap, 2), np.float64)
        _check(L.modsx_shard_device_unpack(self._c(), _p(blocks), int(world), int(items), int(block_rows), int(ndesc), int(row_format),
                                           _p(regs), _ptr_array(descs), _p(pos), C.c_long(cap)), "shard_device_unpack")
        return regs, descs, pos

    def detect_msers(self, img, params=None, tilt=1.0, zoom=1.0):
        params = params or default_mser_params()
        out = C.c_void_p()
        n = _check(lib().modsx_detect_msers(self._c(), C.c_void_p(img.h), C.byref(params), C.c_double(tilt),
                                            C.c_double(zoom), C.byref(out)), "detect_msers")
        return _take(out, n, KEYPOINT)

    def match_ladder(self, img1, img2, steps, params, min_matches=10, comm=None):
        """steps: list of (views, match_ratio[, detector[, descs]]); descs = [(descriptor type, FGINN ratio), ...] is the
        section's Descriptors / FGINNThreshold lists (None: the parameter block's).  Returns (result dict, steps executed).
        comm: a communicator handle => modsx_match_ladder_sharded (every step's views sharded over the ranks)."""
        arr = (LadderStep * len(steps))()
        keep = []
        for i, st in enumerate(steps):
            views, ratio = st[0], st[1]
            va = _view_array(views)
            keep.append(va)
            arr[i].views = C.cast(va, C.c_void_p)
            arr[i].nviews = len(views)
            arr[i].match_ratio = float(ratio)
            arr[i].detector = int(st[2]) if len(st) > 2 else 0
            if len(st) > 3 and st[3] is not None:
                _set_descs(arr[i], st[3])
        res = PairResult()
        done = C.c_int(0)
        if comm is not None:
            _check(lib().modsx_match_ladder_sharded(self._c(), C.c_void_p(comm), C.c_void_p(img1.h), C.c_void_p(img2.h), arr,
                                                    len(steps), int(min_matches), C.byref(params), C.byref(res),
                                                    C.byref(done)), "match_ladder_sharded")
        else:
            _check(lib().modsx_match_ladder(self._c(), C.c_void_p(img1.h), C.c_void_p(img2.h), arr, len(steps),
                                            int(min_matches), C.byref(params), C.byref(res), C.byref(done)), "match_ladder")
        return _unpack_pair_result(res), done.value

    def match_pair(self, img1, img2, params):
        res = PairResult()
        _check(lib().modsx_match_pair(self._c(), C.c_void_p(img1.h), C.c_void_p(img2.h), C.byref(params),
                                      C.byref(res)), "match_pair")
        return _unpack_pair_result(res)

    def last_timings(self):
        t = (C.c_double * 6)()
        _check(lib().modsx_last_timings(self._c(), t), "last_timings")
        return dict(zip(["detect", "orient", "describe", "match", "verify", "total"], list(t)))

    def profile(self, enable=True):
        _check(lib().modsx_profile(self._c(), int(enable)), "profile")

    def kernel_stats(self):
        n = len(KERNEL_CLASSES)
        ms, work = (C.c_double * n)(), (C.c_double * n)()
        launches = (C.c_long * n)()
        _check(lib().modsx_kernel_stats(self._c(), ms, work, launches, n), "kernel_stats")
        return {k: dict(ms=ms[i], work=work[i], launches=launches[i]) for i, k in enumerate(KERNEL_CLASSES)}


class _ImageStruct(C.Structure):   # mirrors struct modsx_image (engine.hpp) for reading rows/cols of a handle
    _fields_ = [("d", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("owned", C.c_bool)]


def last_batch_verify_each():
    """Verification time in ms of every pair of the last batch call (completion order)."""
    n = lib().modsx_debug_last_batch_verify_each(None, 0)
    out = np.zeros(max(n, 1), np.float64)
    n = min(n, lib().modsx_debug_last_batch_verify_each(_p(out), len(out)))
    return out[:n]


def last_batch_verify():
    """(summed ms of DuplicateFiltering + LO-RANSAC, pairs, helper threads) of the last match_pairs call."""
    ms, n, t = C.c_double(), C.c_int(), C.c_int()
    lib().modsx_last_batch_verify(C.byref(ms), C.byref(n), C.byref(t))
    return ms.value, n.value, t.value


def view_block_order(counts):
    """modsx_view_block_order: counts [world, nviews] -> (source row of every list position, maxrows)."""
    counts = np.ascontiguousarray(counts, np.int32)
    world, nviews = counts.shape
    total = int(counts.sum())
    src = np.zeros(max(1, total), np.int32)
    mr = C.c_int(0)
    n = _check(lib().modsx_view_block_order(_p(counts), world, nviews, _p(src), len(src), C.byref(mr)), "view_block_order")
    return src[:n].copy(), mr.value


def shard_owner_plan(item_counts, nviews, world, rank, image_owner):
    """modsx_shard_owner_plan: the owner-only exchange of `rank` from the per-item counts (item f = image * nviews + view) ->
    dict(sends, recvs: [k, 4] = peer, image, first row, rows; jobs: [k, 4] = receive row, list row, rows, 0; recv_rows, list_rows)."""
    cnt = np.ascontiguousarray(item_counts, np.int32)
    own = np.ascontiguousarray(image_owner, np.int32)
    nimg = len(own)
    assert len(cnt) == nimg * nviews
    cap = max(1, len(cnt), nimg * world)
    sends, recvs, jobs = (np.zeros((cap, 4), np.int32) for _ in range(3))
    n = (C.c_long * 5)()
    L = lib()
    L.modsx_shard_owner_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    _check(L.modsx_shard_owner_plan(_p(cnt), nimg, nviews, world, rank, _p(own), _p(sends), _p(recvs), _p(jobs), cap, n), "shard_owner_plan")
    return {"sends": sends[:n[0]].copy(), "recvs": recvs[:n[1]].copy(), "jobs": jobs[:n[2]].copy(), "recv_rows": int(n[3]), "list_rows": int(n[4])}


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def shard_region_bytes(row_format):
    return REGION.itemsize if row_format == SHARD_ROW_REGION else 56


def shard_kp_rows(regs):
    """The SHARD_ROW_KP slice of a region array: [n, 7] f64 = x, y, a11, a12, a21, a22, s of reproj_kp."""
    k = np.asarray(regs)["reproj_kp"]
    return np.stack([k[f] for f in KP_FIELDS], 1).astype(np.float64) if len(k) else np.zeros((0, 7))


def shard_block_bytes(items, block_rows, ndesc=1, row_format=SHARD_ROW_REGION):
    L = lib()
    L.modsx_shard_block_bytes.restype = C.c_long
    return int(L.modsx_shard_block_bytes(int(items), int(block_rows), int(ndesc), int(row_format)))


def shard_block_pack(regs, descs, counts, block_rows, rc_local=0, row_format=SHARD_ROW_REGION):
    """This rank's block of one exchange (host statement of the wire format): regs in item order, descs = list of [n, 128] u8
    arrays (one per descriptor class), counts[f] per (image, view) item.  Returns the block as a uint8 array."""
    L = lib()
    L.modsx_shard_block_pack.restype = C.c_long
    regs = np.ascontiguousarray(regs, REGION)
    descs = [np.ascontiguousarray(d, np.uint8) for d in descs]
    counts = np.ascontiguousarray(counts, np.int32)
    out = np.zeros(shard_block_bytes(len(counts), block_rows, len(descs), row_format), np.uint8)
    n = _check(L.modsx_shard_block_pack(_p(regs), _ptr_array(descs), len(descs), len(regs), _p(counts), len(counts), int(rc_local),
                                        int(block_rows), int(row_format), _p(out)), "shard_block_pack")
    assert n == len(out)
    return out


def shard_blocks_unpack(blocks, world, items, block_rows, ndesc=1, cap=None, row_format=SHARD_ROW_REGION):
    """The reference's list from `world` gathered blocks (host): (regs -- REGION records, or [n, 7] f64 for SHARD_ROW_KP --,
    [desc per class], item_counts).  Raises on a failed rank; returns (None, None, need_rows) when a block was too small."""
    L = lib()
    L.modsx_shard_blocks_unpack.restype = C.c_long
    blocks = np.ascontiguousarray(blocks, np.uint8)
    cap = int(cap if cap is not None else world * block_rows)
    regs = np.zeros(max(1, cap), REGION) if row_format == SHARD_ROW_REGION else np.zeros((max(1, cap), 7), np.float64)
    descs = [np.zeros((max(1, cap), 128), np.uint8) for _ in range(ndesc)]
    cnt = np.zeros(items, np.int32)
    need = C.c_int(0)
    n = L.modsx_shard_blocks_unpack(_p(blocks), int(world), int(items), int(block_rows), int(ndesc), int(row_format), _p(regs),
                                    _ptr_array(descs), C.c_long(cap), _p(cnt), C.byref(need))
    if n == -5:      # MODSX_ERR_CAPACITY
        return None, None, need.value
    _check(n, "shard_blocks_unpack")
    return regs[:n].copy(), [d[:n].copy() for d in descs], cnt


def comm_unique_id():
    buf = C.create_string_buffer(128)
    _check(lib().modsx_comm_unique_id(buf), "comm_unique_id")
    return buf.raw


def comm_loopback_id(world):
    """Id of an in-process communicator group: `world` ranks of THIS process on one device (one context + thread each)."""
    buf = C.create_string_buffer(128)
    _check(lib().modsx_comm_loopback_id(buf, int(world)), "comm_loopback_id")
    return buf.raw


COMM_STATS = ["collectives", "bytes_gathered", "block_retries", "agreements", "lanes", "loopback", "dead", "turn_wait_us", "bytes_received",
              "exchanges"]


def comm_stats(comm):
    out = (C.c_long * len(COMM_STATS))()
    lib().modsx_comm_stats(C.c_void_p(comm), out, len(COMM_STATS))
    return dict(zip(COMM_STATS, list(out)))


def _image_dims(handle):
    st = _ImageStruct.from_address(handle)
    return st.rows, st.cols


def match_pairs(ctxs, imgs1, imgs2, params):
    """modsx_match_pairs: a batch of pairs pipelined over several contexts (threads + streams)."""
    n = len(imgs1)
    carr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    a1 = (C.c_void_p * n)(*[im.h for im in imgs1])
    a2 = (C.c_void_p * n)(*[im.h for im in imgs2])
    res = (PairResult * n)()
    _check(lib().modsx_match_pairs(carr, len(ctxs), a1, a2, n, C.byref(params), res), "match_pairs")
    return [_unpack_pair_result(res[i]) for i in range(n)]


def match_pairs_views(ctxs, imgs1, imgs2, views, params, arrays=True):
    """modsx_match_pairs_views: a batch of multi-view pairs over several contexts, verification on helper threads.
    arrays=False returns the counts and H only (the tentative / flag arrays are released without being copied out)."""
    n = len(imgs1)
    carr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    a1 = (C.c_void_p * n)(*[im.h for im in imgs1])
    a2 = (C.c_void_p * n)(*[im.h for im in imgs2])
    arr = _view_array(views)
    res = (PairResult * n)()
    _check(lib().modsx_match_pairs_views(carr, len(ctxs), a1, a2, n, arr, len(views), C.byref(params), res), "match_pairs_views")
    return [_unpack_pair_result(res[i], arrays) for i in range(n)]


def _unpack_pair_result(res, arrays=True):
    if True:
        T = res.n_unique if arrays else 0
        out = dict(n_regions=(res.n_regions1, res.n_regions2), n_tentatives=res.n_tentatives, n_unique=res.n_unique,
                   n_ransac_inliers=res.n_ransac_inliers, n_verified=res.n_verified,
                   ransac_samples=res.ransac_samples, ransac_lo=res.ransac_lo,
                   H=np.array(list(res.H)).reshape(3, 3))
        if T > 0:
            out["tentatives"] = np.frombuffer((C.c_char * (T * TENT.itemsize)).from_address(res.tentatives),
                                              dtype=TENT, count=T).copy()
            out["ransac_inlier"] = np.frombuffer((C.c_char * T).from_address(res.ransac_inlier), np.uint8, T).astype(bool)
            out["verified"] = np.frombuffer((C.c_char * T).from_address(res.verified), np.uint8, T).astype(bool)
        else:
            out["tentatives"] = np.zeros(0, TENT)
            out["ransac_inlier"] = np.zeros(0, bool)
            out["verified"] = np.zeros(0, bool)
        lib().modsx_pair_result_release(C.byref(res))
        return out

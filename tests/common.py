"""Shared helpers for the parity tests (oracle-side pipelines, synthetic correspondences)."""
import numpy as np


def oracle_features(O, g, mode=0, rootsift=1, **kw):
    p = O.default_params(mode=mode, **kw)
    k = O.detect_hessaff(g, p)
    r = O.detect_affine_regions(k)
    ro = O.detect_orientation(g, r)
    rr = O.reproject_regions(ro, np.eye(3), g.shape[1], g.shape[0])
    d = O.describe_regions(g, rr, rootsift=rootsift)
    return k, rr, d


def laf_of(regs, idx):
    k = regs["reproj_kp"][idx]
    return np.stack([k["a11"], k["a12"], k["a21"], k["a22"], k["s"]], 1)


def oracle_pair(O, a, b, seed=1, mode=0, ratio=0.8, contrad=30.0, **det_kw):
    k1, r1, d1 = oracle_features(O, a, mode, **det_kw)
    k2, r2, d2 = oracle_features(O, b, mode, **det_kw)
    pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
    tent = O.match_fginn(d1, d2, pos2, ratio, contrad)
    pts = np.stack([r1["reproj_kp"]["x"][tent["q"]], r1["reproj_kp"]["y"][tent["q"]],
                    r2["reproj_kp"]["x"][tent["t0"]], r2["reproj_kp"]["y"][tent["t0"]]], 1)
    order, keep = O.duplicate_filtering(pts, tent["ratio"], 2.0, True)
    sel = order[keep]
    tu, pu = tent[sel], pts[sel]
    res = O.loransac_h(pu, laf_of(r1, tu["q"]), laf_of(r2, tu["t0"]), seed=seed) if O.ref_available() else None
    return dict(r1=r1, r2=r2, d1=d1, d2=d2, tent=tent, uniq=tu, pts=pu, ransac=res)


def synth_corr(T, frac, noise=0.7, seed=0):
    rs = np.random.RandomState(seed)
    H = np.array([[1.1, 0.05, 20], [-0.03, 0.95, -10], [1e-5, 2e-5, 1]])
    p1 = rs.uniform(20, 1000, (T, 2))
    x = np.c_[p1, np.ones(T)] @ H.T
    p2 = x[:, :2] / x[:, 2:] + rs.normal(0, noise, (T, 2))
    out = rs.rand(T) > frac
    p2[out] = rs.uniform(20, 1000, (int(out.sum()), 2))
    laf = np.tile([1, 0, 0, 1, 5.0], (T, 1)).astype(float)
    return np.c_[p1, p2], laf, H


def synth_two_view(seed, n_in=300, n_out=150, planar_frac=0.0, noise=0.5):
    """Correspondences of a 3-D point cloud seen by two cameras (+ uniform outliers).  planar_frac of the inliers lie
    on one plane, which drives exp_ransacFcustom through its DEGENSAC (plane-and-parallax) branch."""
    rng = np.random.default_rng(seed)
    K = np.array([[800., 0, 512], [0, 800, 384], [0, 0, 1]])
    X = np.stack([rng.uniform(-4, 4, n_in), rng.uniform(-3, 3, n_in), rng.uniform(6, 14, n_in)], 1)
    npl = int(planar_frac * n_in)
    if npl:
        X[:npl, 2] = 10.0 + 0.15 * X[:npl, 0]
    ang = 0.25
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([-1.5, 0.1, 0.3])
    p1 = (K @ X.T).T
    p1 = p1[:, :2] / p1[:, 2:]
    X2 = (R @ X.T).T + t
    p2 = (K @ X2.T).T
    p2 = p2[:, :2] / p2[:, 2:]
    p1 += rng.normal(0, noise, p1.shape)
    p2 += rng.normal(0, noise, p2.shape)
    o1 = np.stack([rng.uniform(0, 1024, n_out), rng.uniform(0, 768, n_out)], 1)
    o2 = np.stack([rng.uniform(0, 1024, n_out), rng.uniform(0, 768, n_out)], 1)
    pts = np.concatenate([np.concatenate([p1, p2], 1), np.concatenate([o1, o2], 1)])
    pts = pts[rng.permutation(len(pts))]
    laf = np.tile(np.array([1., 0, 0, 1, 3.0]), (len(pts), 1))
    return pts, laf


def normH(H):
    H = np.asarray(H, float).reshape(3, 3)
    return H / H[2, 2]


def same_records(a, b):
    """Field-wise equality of two structured arrays (ignores struct padding bytes)."""
    if len(a) != len(b):
        return False
    for name in a.dtype.names:
        x, y = a[name], b[name]
        if x.dtype.names:
            if not same_records(x, y):
                return False
        elif not np.array_equal(x, y):
            return False
    return True


def dev_to_host(ptr, nbytes):
    """Copy nbytes from a device pointer (e.g. the HBM-resident u8 descriptors of a sharded call) into a numpy array."""
    import ctypes as C
    import numpy as np
    hip = C.CDLL("libamdhip64.so")
    out = np.empty(int(nbytes), np.uint8)
    hip.hipDeviceSynchronize()
    rc = hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(int(nbytes)), 2)
    assert rc == 0, rc
    return out


DESC_NAME_ORDER = (3, 2, 1, 0)   # "HalfRootSIFT" < "HalfSIFT" < "RootSIFT" < "SIFT": the outer key of CorrespondencesMapMap
DET_NAME_ORDER = (0, 3)          # "HessianAffine" < "MSER"


def oracle_ladder(oracle, a, b, steps, min_matches, seed, ori=(1.0, 41, 1, 0.8), threads=1, hess=None, mser_kw=None,
                  contrad=30.0, dup_dist=2.0, ransac=None, default_desc=1):
    """mods.cpp:229-415 restated with the oracle's stage functions (test-side only).
    steps: (views, ratio[, detector[, descs]]) with descs = [(descriptor type, FGINN ratio), ...] -- the Descriptors /
    FGINNThreshold lists of the step's [DetectorN] section (default: one class {default_desc, ratio}).
    Every (detector, descriptor) class keeps its own region lists (RegionVectorMap[det][desc], imagerepresentation.cpp:552-600)
    and tentatives (CorrespondencesMapMap[desc][det], correspondencebank.cpp:180-218); a step orients once (Half-folded iff
    one of its descriptors is a Half type) and describes all its descriptors on that list; MatchImgReps re-matches the classes
    of the step's detector with the step's thresholds (:291-347); GetCorresponcesVector concatenates in map order (:117-179).
    ransac: dict(kind="h"|"f", **kwargs of oracle.loransac_h / loransac_f)."""
    mser_kw = mser_kw or dict(min_size=30, max_area=0.05, min_margin=8.0)
    ransac = dict(ransac or dict(kind="h"))
    kind = ransac.pop("kind", "h")
    cls = {}   # (desc type, det) -> dict(acc=[[regs, desc], [regs, desc]], tent)
    out, done, cur = None, 0, 0
    for st in steps:
        if cur >= min_matches:
            break
        views, ratio = st[0], st[1]
        det = st[2] if len(st) > 2 else 0
        descs = st[3] if len(st) > 3 and st[3] is not None else [(default_desc, ratio)]
        types = [t for t, _ in descs]
        for side, img in enumerate((a, b)):
            r, ds = oracle.detect_describe_views(img, views, params=hess, ori=ori, mser=mser_kw if det == 3 else None,
                                                 threads=threads, descs=types)
            for t, d in zip(types, ds):
                k = cls.setdefault((t, det), dict(acc=[[None, None], [None, None]], tent=None))
                if k["acc"][side][0] is None:
                    k["acc"][side] = [r.copy(), d]
                else:
                    rr = r.copy()
                    rr["id"] += len(k["acc"][side][0]); rr["parent_id"] += len(k["acc"][side][0])     # AddRegionsToList
                    k["acc"][side] = [np.concatenate([k["acc"][side][0], rr]), np.concatenate([k["acc"][side][1], d])]
        for t, thr in descs:
            k = cls[(t, det)]
            (r1, d1), (r2, d2) = k["acc"]
            pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
            k["tent"] = oracle.match_fginn(d1, d2, pos2, thr, contrad)
        R1, R2, T = [], [], []
        o1 = o2 = 0
        for t in DESC_NAME_ORDER:
            for dkey in DET_NAME_ORDER:
                kk = cls.get((t, dkey))
                if kk is None:
                    continue
                tt = kk["tent"].copy()
                tt["q"] += o1
                for f in ("t0", "t1", "tj"):
                    tt[f] = np.where(tt[f] >= 0, tt[f] + o2, tt[f])
                T.append(tt); R1.append(kk["acc"][0][0]); R2.append(kk["acc"][1][0])
                o1 += len(kk["acc"][0][0]); o2 += len(kk["acc"][1][0])
        r1, r2, tent = np.concatenate(R1), np.concatenate(R2), np.concatenate(T)
        pts = np.stack([r1["reproj_kp"]["x"][tent["q"]], r1["reproj_kp"]["y"][tent["q"]],
                        r2["reproj_kp"]["x"][tent["t0"]], r2["reproj_kp"]["y"][tent["t0"]]], 1)
        order, keep = oracle.duplicate_filtering(pts, tent["ratio"], dup_dist, True)
        sel = order[keep]
        tu, pu = tent[sel], pts[sel]
        if kind == "f":
            rr = oracle.loransac_f(pu, laf_of(r1, tu["q"]), laf_of(r2, tu["t0"]), seed=seed, **ransac)
        else:
            rr = oracle.loransac_h(pu, laf_of(r1, tu["q"]), laf_of(r2, tu["t0"]), seed=seed, **ransac)
        cur = int(rr["keep"].sum())
        out = dict(n_regions=(len(r1), len(r2)), n_tentatives=len(tent), tent=tu, rr=rr, r1=r1, r2=r2, pts=pu)
        done += 1
    return out, done


def need_ref(oracle):
    """The RANSAC comparisons are made against the reference's own degensac (oracle/_ref, built from /root/reference in place and
    carried to the GPU box with the snapshot).  On a GPU box a missing build is an ERROR, not a skip: a green run must mean that
    the comparisons were made."""
    assert oracle.ref_available(), "oracle/_ref is not built: run `make -C oracle ref` where /root/reference exists"

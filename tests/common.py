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


def oracle_pair(O, a, b, seed=1, mode=0, ratio=0.8, contrad=30.0):
    k1, r1, d1 = oracle_features(O, a, mode)
    k2, r2, d2 = oracle_features(O, b, mode)
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

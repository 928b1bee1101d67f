"""Seeded synthetic image pairs.

Image A = 128 + sum of anisotropic Gaussian blobs + uniform noise, floored and
clamped to [0,255]; image B = A warped by a fixed homography (bilinear) plus
independent noise.  Pure numpy, used by tests/, bench.py and smoke() to make
inputs; it is not part of the device path.
"""
import numpy as np

H_DEFAULT = np.array([[1.1, 0.05, 20.0], [-0.03, 0.95, -10.0], [1e-5, 2e-5, 1.0]])


def blob_image(rows=768, cols=1024, nblobs=4000, seed=12345):
    rs = np.random.RandomState(seed)
    img = np.full((rows, cols), 128.0, np.float64)
    cx = rs.uniform(0, cols, nblobs)
    cy = rs.uniform(0, rows, nblobs)
    sig = rs.uniform(1.5, 13.5, nblobs)
    amp = rs.uniform(-80, 80, nblobs)
    ecc = rs.uniform(0.5, 1.5, nblobs)
    ang = rs.uniform(0, np.pi, nblobs)
    for i in range(nblobs):
        sx, sy = sig[i], sig[i] * ecc[i]
        rad = int(4 * max(sx, sy)) + 1
        x0, x1 = max(0, int(cx[i]) - rad), min(cols, int(cx[i]) + rad + 1)
        y0, y1 = max(0, int(cy[i]) - rad), min(rows, int(cy[i]) + rad + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        xs = np.arange(x0, x1) - cx[i]
        ys = np.arange(y0, y1) - cy[i]
        X, Y = np.meshgrid(xs, ys)
        c, s = np.cos(ang[i]), np.sin(ang[i])
        u = c * X + s * Y
        v = -s * X + c * Y
        img[y0:y1, x0:x1] += amp[i] * np.exp(-0.5 * ((u / sx) ** 2 + (v / sy) ** 2))
    img += rs.uniform(-2, 2, img.shape)
    return np.clip(np.floor(img), 0, 255).astype(np.float32)


def warp_homography(img, H, seed=54321, noise=2.0):
    """B(x') = A(H^-1 x') with bilinear sampling, 128 outside, + U[-noise,noise], floored."""
    rows, cols = img.shape
    Hi = np.linalg.inv(H)
    xs, ys = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    w = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    sx = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / w
    sy = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / w
    x0 = np.floor(sx).astype(np.int64)
    y0 = np.floor(sy).astype(np.int64)
    fx, fy = sx - x0, sy - y0
    ok = (x0 >= 0) & (y0 >= 0) & (x0 < cols - 1) & (y0 < rows - 1)
    x0c, y0c = np.clip(x0, 0, cols - 2), np.clip(y0, 0, rows - 2)
    a = img.astype(np.float64)
    v = ((1 - fy) * ((1 - fx) * a[y0c, x0c] + fx * a[y0c, x0c + 1]) +
         fy * ((1 - fx) * a[y0c + 1, x0c] + fx * a[y0c + 1, x0c + 1]))
    v = np.where(ok, v, 128.0)
    rs = np.random.RandomState(seed)
    v = v + rs.uniform(-noise, noise, v.shape)
    return np.clip(np.floor(v), 0, 255).astype(np.float32)


def make_pair(rows=768, cols=1024, nblobs=4000, seed=12345, H=None):
    H = H_DEFAULT if H is None else np.asarray(H, np.float64)
    a = blob_image(rows, cols, nblobs, seed)
    b = warp_homography(a, H, seed=seed + 42000)
    return a, b, H


# ---- many pairs at once: a cache of generated images + worker processes ---------------------------------------------------------
# bench.py runs every configuration on as many DISTINCT pairs as a step holds (64 at 1024x768, 256 at 1920x1080 for configs[4]);
# the blob renderer above is a python loop (~1 s per 1024x768 pair on the GPU box's host cores), so the pairs are rendered by worker
# processes and kept as u8 files (the images are floored and clipped: integers 0..255, so u8 is exact) under a cache directory that
# later runs on the same box reuse.  spec = (rows, cols, nblobs, seed_a, seed_b): image A = blob_image(seed_a), image B = A warped by
# H_DEFAULT with noise seed seed_b -- make_pair(seed) is the spec (.., seed, seed + 42000).
def _cache_dir():
    import os
    d = os.environ.get("MODSX_SYNTH_CACHE", "/tmp/modsx_synth_cache")
    try:
        os.makedirs(d, exist_ok=True)
        return d if os.access(d, os.W_OK) else None
    except OSError:
        return None


def _spec_path(d, spec):
    import os
    return os.path.join(d, "pair_%dx%d_b%d_a%d_b%d.npy" % tuple(spec))


def _render(spec):
    rows, cols, nblobs, sa, sb = spec
    a = blob_image(rows, cols, nblobs, sa)
    b = warp_homography(a, H_DEFAULT, seed=sb)
    return np.stack([a, b]).astype(np.uint8)


def _render_to_cache(d, spec):
    import os
    path = _spec_path(d, spec)
    tmp = path + ".tmp%d.npy" % os.getpid()
    np.save(tmp, _render(spec))
    os.replace(tmp, path)


def make_pairs(specs, procs=1, as_u8=False):
    """-> [(a, b, H)] for specs = [(rows, cols, nblobs, seed_a, seed_b)], f32 images (u8 with as_u8) -- the arrays make_pair gives."""
    import os
    import subprocess
    import sys
    import json
    specs = [tuple(int(x) for x in s) for s in specs]
    d = _cache_dir()
    out = {}
    if d is not None:
        missing = [s for s in dict.fromkeys(specs) if not os.path.exists(_spec_path(d, s))]
        procs = max(1, min(int(procs), len(missing)))
        if missing and procs > 1:
            children = [subprocess.Popen([sys.executable, os.path.abspath(__file__), d, json.dumps(missing[i::procs])],
                                         env=dict(os.environ, OMP_NUM_THREADS="1")) for i in range(procs)]
            for c in children:
                c.wait()
        for s in dict.fromkeys(specs):
            path = _spec_path(d, s)
            if not os.path.exists(path):
                _render_to_cache(d, s)      # one process, or a child that failed
            try:
                out[s] = np.load(path)
            except (OSError, ValueError):
                out[s] = _render(s)
    else:
        for s in dict.fromkeys(specs):
            out[s] = _render(s)
    res = []
    for s in specs:
        ab = out[s]
        res.append((ab[0], ab[1], H_DEFAULT) if as_u8 else (ab[0].astype(np.float32), ab[1].astype(np.float32), H_DEFAULT))
    return res


if __name__ == "__main__":      # worker: synthetic.py <cache dir> <json list of specs>
    import json
    import sys
    for _s in json.loads(sys.argv[2]):
        _render_to_cache(sys.argv[1], tuple(_s))

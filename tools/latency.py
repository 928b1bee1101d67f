import os as _os
_os.environ.setdefault("MODSX_MALLOC_TUNE", "1")   # modsx.h: opt-in allocator tuning
import sys, time, numpy as np
sys.path.insert(0, '.')
import mods_amd
from mods_amd import synthetic
ctx = mods_amd.Context(0)
a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
ia, ib = ctx.upload(a), ctx.upload(b)
par = mods_amd.default_pair_params(ransac_seed=1)
for name, views in (("1 view", None), ("8 views", mods_amd.set_vs_pars([1.0], [1,2,3,4,6], 360.0, 0.2, 1, [])), ("31 views", mods_amd.set_vs_pars([1.0], [1,2,4,6,8], 120.0, 0.2, 1, []))):
    f = (lambda: ctx.match_pair(ia, ib, par)) if views is None else (lambda: ctx.match_pair_views(ia, ib, views, par))
    for _ in range(4): f()            # helper contexts and their buffers are created over the first calls
    ts = []
    for _ in range(25):
        t = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t)
    dt = float(np.median(ts))         # median of 25: a lone pair shows an occasional +5..10 ms call (box dependent)
    ctx.profile(True); f(); st = ctx.kernel_stats(); ctx.profile(False)
    gpu = sum(v["ms"] for v in st.values())
    print("%s: %.2f ms wall per pair (median of 25; mean %.2f; one context, idle GPU), %.2f ms of kernels, %d launches; stages %s" % (name, dt * 1e3, float(np.mean(ts)) * 1e3, gpu, sum(v["launches"] for v in st.values()), ctx.last_timings() if views is None else ""))

"""One 31-view pair repeated: the distribution of its wall time (single context, idle GPU)."""
import os as _os
_os.environ.setdefault("MODSX_MALLOC_TUNE", "1")   # modsx.h: opt-in allocator tuning
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import mods_amd
from mods_amd import synthetic
ctx = mods_amd.Context(0)
a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
ia, ib = ctx.upload(a), ctx.upload(b)
par = mods_amd.default_pair_params(ransac_seed=1)
views = mods_amd.set_vs_pars([1.0], [1, 2, 4, 6, 8], 120.0, 0.2, 1, [])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for _ in range(3):
    ctx.match_pair_views(ia, ib, views, par)
ts = []
for _ in range(n):
    t = time.perf_counter(); ctx.match_pair_views(ia, ib, views, par); ts.append((time.perf_counter() - t) * 1e3)
ts = np.array(ts)
print("n %d: min %.2f median %.2f mean %.2f p90 %.2f max %.2f ms" % (n, ts.min(), np.median(ts), ts.mean(), np.percentile(ts, 90), ts.max()))
print(" ".join("%.1f" % t for t in ts))

"""One 31-view pair, repeated: for kernel-trace timelines of the single-pair (reference CLI) use."""
import os as _os
_os.environ.setdefault("MODSX_MALLOC_TUNE", "1")   # modsx.h: opt-in allocator tuning
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mods_amd
from mods_amd import synthetic
ctx = mods_amd.Context(0)
a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
ia, ib = ctx.upload(a), ctx.upload(b)
par = mods_amd.default_pair_params(ransac_seed=1)
views = mods_amd.set_vs_pars([1.0], [1, 2, 4, 6, 8], 120.0, 0.2, 1, [])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for _ in range(2):
    ctx.match_pair_views(ia, ib, views, par)
t = time.perf_counter()
for _ in range(n):
    ctx.match_pair_views(ia, ib, views, par)
print("%.2f ms per pair" % ((time.perf_counter() - t) / n * 1e3))

import sys, time, numpy as np
sys.path.insert(0, '.')
import mods_amd
from mods_amd import synthetic, distributed as D
a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
ctx = mods_amd.Context(0)
ia, ib = ctx.upload(a), ctx.upload(b)
par = mods_amd.default_pair_params(ransac_seed=1)
views = mods_amd.set_vs_pars([1.0], [1,2,4,6,8], 120.0, 0.2, 1, [])
ref = ctx.match_pair_views(ia, ib, views, par)
for W in (1, 2, 4, 8):
    def body(r, comm):
        c = comm.ctxs[0]
        for _ in range(3):
            c.match_pair_views_sharded(comm.comm, ia, ib, views, par, 0)
        t = time.perf_counter(); n = 8
        for _ in range(n):
            res = c.match_pair_views_sharded(comm.comm, ia, ib, views, par, 0)
        return (time.perf_counter() - t) / n, res
    out = D.run_loopback(W, body)
    ok = out[0][1]["n_verified"] == ref["n_verified"] and np.array_equal(out[0][1]["H"], ref["H"])
    print("loopback W=%d: %.2f ms per 31-view pair (rank 0 verifies), identical to unsharded: %s" % (W, out[0][0] * 1e3, ok))

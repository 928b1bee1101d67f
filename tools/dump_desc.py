import sys, numpy as np
sys.path.insert(0, '.')
import mods_amd
from mods_amd import synthetic
ctx = mods_amd.Context(0)
a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
views = mods_amd.set_vs_pars([1.0], [1,2,4,6,8], 120.0, 0.2, 1, [])
par = mods_amd.default_pair_params()
ia, ib = ctx.upload(a), ctx.upload(b)
r1, d1 = ctx.detect_describe_views(ia, views, par)
r2, d2 = ctx.detect_describe_views(ib, views, par)
pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
np.savez_compressed("gpurun_out/r3/desc31.npz", d1=d1.astype(np.uint8), d2=d2.astype(np.uint8), pos2=pos2)
print(d1.shape, d2.shape)

"""Dump the verifier's inputs of one 31-view pair (tentative coordinates, ratios, LAFs) for host-side profiling off the GPU box."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import mods_amd
from mods_amd import synthetic
ctx = mods_amd.Context(0)
a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
views = mods_amd.set_vs_pars([1.0], [1, 2, 4, 6, 8], 120.0, 0.2, 1, [])
par = mods_amd.default_pair_params(ransac_seed=1)
ia, ib = ctx.upload(a), ctx.upload(b)
r1, d1 = ctx.detect_describe_views(ia, views, par)
r2, d2 = ctx.detect_describe_views(ib, views, par)
pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
t = ctx.match_fginn(d1, d2, pos2)
k1, k2 = r1["reproj_kp"][t["q"]], r2["reproj_kp"][t["t0"]]
pts = np.stack([k1["x"], k1["y"], k2["x"], k2["y"]], 1)
laf = lambda k: np.stack([k["a11"], k["a12"], k["a21"], k["a22"], k["s"]], 1)
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3h/tents31.npz"
np.savez_compressed(out, pts=pts, key=t["ratio"], laf1=laf(k1), laf2=laf(k2))
print(len(t), "tentatives ->", out)

import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, mods_amd
from mods_amd import synthetic
import bench
a, b, _ = synthetic.make_pair(rows=1080, cols=1920, nblobs=int(4000 * 1920 * 1080 / (1024 * 768)), seed=12345)
ctx = mods_amd.Context(0)
p = mods_amd.default_pair_params(ransac_seed=1, **bench.WXBS)
ia, ib = ctx.upload(a), ctx.upload(b)
for i in range(3):
    r = mods_amd.match_pairs([ctx], [ia], [ib], p)
os.environ["MODSX_HOST_TIMING"] = "2"
t0 = time.time(); r = mods_amd.match_pairs([ctx], [ia], [ib], p); print("pair %.1f ms" % (1e3 * (time.time() - t0)), r[0]["n_regions"], r[0]["n_tentatives"], file=sys.stderr)

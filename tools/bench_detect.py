"""Detection-only driver for kernel experiments: one 1024x768 synthetic image, a 5-view set, detect (+ optionally orient /
describe) `--reps` times.  Prints the region count; meant to be run under rocprofv3 (tools/prof_cmd.sh)."""
import argparse, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mods_amd
from mods_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tilts", type=str, default="1,2,4,6,8")
ap.add_argument("--phi", type=float, default=120.0)
ap.add_argument("--desc", type=int, default=0)
args = ap.parse_args()
ctx = mods_amd.Context(0)
a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
ia = ctx.upload(a)
views = mods_amd.set_vs_pars([1.0], [float(t) for t in args.tilts.split(",")], args.phi, 0.2, 1, [])
for _ in range(args.reps):
    r, d = ctx.detect_describe_views(ia, views, mods_amd.default_pair_params(), want_desc=bool(args.desc))
print(len(views), "views", len(r), "regions")

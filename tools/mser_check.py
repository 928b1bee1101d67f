#!/usr/bin/env python3
"""CPU-only: the host MSER (mser.cpp behind modsx_detect_msers_u8) against the oracle's independent restatement (oracle_mser.cpp)
on random images of random sizes and parameters -- key for key, bit for bit -- and the time of a 1024x768 view.
usage: mser_check.py [images] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import mods_amd
from mods_amd import synthetic
import pyoracle as O
from common import same_records

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
bad = 0
for i in range(n):
    rows, cols = int(rs.randint(40, 500)), int(rs.randint(40, 640))
    a, _, _ = synthetic.make_pair(rows=rows, cols=cols, nblobs=int(rs.randint(5, 400)), seed=int(rs.randint(1, 10 ** 6)))
    kind = rs.randint(0, 4)
    if kind == 1: a = a + rs.normal(0, 12, a.shape)                       # noise: many tiny components
    if kind == 2: a = np.round(a / 32) * 32                              # plateaus: many pixels per level
    if kind == 3: a = rs.randint(0, 256, a.shape).astype(np.float64)     # pure noise
    g = np.clip(a, 0, 255).astype(np.uint8)
    kw = dict(min_size=int(rs.choice([5, 30, 60])), max_area=float(rs.choice([0.01, 0.05, 0.5])), min_margin=float(rs.choice([4, 8, 10, 20])),
              relative=int(rs.choice([0, 0, 1])))
    p = mods_amd.default_mser_params(**kw)
    tilt, zoom = float(rs.choice([1.0, 2.0])), float(rs.choice([1.0, 0.5]))
    k1 = mods_amd.detect_msers_u8(g, p, tilt=tilt, zoom=zoom)
    k2 = O.detect_msers(g.astype(np.float32), mode=int(p.mode), reg_number=int(p.reg_number), rel_threshold=float(p.rel_threshold),
                        rel_reg_number=float(p.rel_reg_number), tilt=tilt, zoom=zoom, **kw)
    if not same_records(k1, k2):
        bad += 1
        print("  image %d (%dx%d kind %d %s): %d keys against %d" % (i, rows, cols, kind, kw, len(k1), len(k2)))
a, _, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
g = np.clip(a, 0, 255).astype(np.uint8)
mods_amd.detect_msers_u8(g)
t0 = time.time()
for _ in range(5):
    k = mods_amd.detect_msers_u8(g)
print("mser_check: %d images, %d differ from the oracle; 1024x768 view (both polarities): %.1f ms, %d keys" % (n, bad, 1e3 * (time.time() - t0) / 5, len(k)))

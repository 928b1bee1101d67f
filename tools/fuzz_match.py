#!/usr/bin/env python3
"""Randomised parity sweep of the device matcher against the oracle's MatchFlannFGINN (GPU box): random sizes from one train to
several thousand (one split to many, empty parity classes, ragged tiles), descriptor alphabets from 2 values (ties everywhere,
zero distances) to SIFT-like, planted runs of near-duplicates at one place (long walks, logged groups, streams that run out of
slots), nn from 2 to 256, ratios below and at / above 1 (the all-points branch).  Every tentative field must agree.
  python tools/fuzz_match.py [n_cases] [seed0]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mods_amd  # noqa: E402
from oracle import pyoracle as O  # noqa: E402


def same(a, b):
    if len(a) != len(b):
        return False
    for f in a.dtype.names:
        x, y = a[f], b[f]
        if x.dtype.kind == "f":
            if not np.array_equal(x, y, equal_nan=True):
                return False
        elif not np.array_equal(x, y):
            return False
    return True


def main(n, seed0):
    O.lib()
    ctx = mods_amd.Context(0)
    rs = np.random.RandomState(seed0)
    bad, t0, nq = 0, time.time(), 0
    for i in range(n):
        kind = rs.randint(4)
        n1 = int(rs.choice([1, 3, 31, 33, 100, 257, 600, 1500]))
        n2 = int(rs.choice([1, 2, 17, 32, 33, 95, 300, 1000, 2500, 6000]))
        if kind == 0:      # tiny alphabet: exact ties, duplicates, d = 0
            lv = int(rs.randint(2, 5))
            d1 = (rs.randint(0, lv, (n1, 128)) * rs.randint(1, 60)).astype(np.float32)
            d2 = (rs.randint(0, lv, (n2, 128)) * rs.randint(1, 60)).astype(np.float32)
            if n2 > 4:
                d2[n2 // 2:] = d2[: n2 - n2 // 2]
        elif kind == 1:    # uniform bytes
            hi = int(rs.choice([16, 90, 256]))
            d1 = rs.randint(0, hi, (n1, 128)).astype(np.float32)
            d2 = rs.randint(0, hi, (n2, 128)).astype(np.float32)
        else:              # SIFT-like, with planted runs of near-duplicates
            def mk(m):
                d = rs.gamma(0.6, 30.0, (m, 128))
                d = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9) * 512
                return np.clip(np.floor(d), 0, 255).astype(np.float32)
            d1, d2 = mk(n1), mk(n2)
        d1 = np.clip(d1, 0, 255); d2 = np.clip(d2, 0, 255)
        pos2 = rs.uniform(0, float(rs.choice([40, 400, 3000])), (n2, 2))
        if kind >= 2 and n2 > 8:
            for q in range(0, n1, int(rs.randint(1, 4))):
                k = int(rs.randint(2, min(n2 - 1, int(rs.choice([4, 20, 80, 300])))))
                st = int(rs.randint(0, n2 - k))
                d2[st:st + k] = np.clip(d1[q][None, :] + rs.randint(-2, 3, (k, 128)), 0, 255)
                if rs.rand() < 0.7:
                    pos2[st:st + k] = pos2[st] + rs.uniform(-3, 3, (k, 2))
                if rs.rand() < 0.3:
                    d2[st + k - 1] = d2[st]
        ratio = float(rs.choice([0.6, 0.8, 0.9, 0.99, 1.0, 1.3]))
        cd = float(rs.choice([2.0, 10.0, 30.0, 1e4]))
        nn = int(rs.choice([2, 3, 8, 20, 50, 65, 256]))
        ref = O.match_fginn(d1, d2, pos2, ratio, cd, nn)
        got = ctx.match_fginn(d1, d2, pos2, ratio, cd, nn)
        nq += n1
        if not same(got, ref):
            bad += 1
            print("MISMATCH case %d: kind %d n1 %d n2 %d ratio %g cd %g nn %d: %d vs %d tentatives" % (i, kind, n1, n2, ratio, cd, nn, len(got), len(ref)), flush=True)
            if bad <= 3:
                np.savez("/tmp/fuzz_match_bad_%d.npz" % i, d1=d1, d2=d2, pos2=pos2, ratio=ratio, cd=cd, nn=nn)
    print("fuzz_match: %d cases (%d queries), %d mismatches, %.1f s" % (n, nq, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)

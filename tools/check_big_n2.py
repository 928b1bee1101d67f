#!/usr/bin/env python3
"""70 000 trains (274 pack workgroups: more than the 256 slots of a waiting workgroup's miss list) against the oracle, both branches of the walk;
with a -DMODSX_PACK_POLLS=0 build (MODSX_LIB=...) every pack workgroup counts all its predecessors itself -- the path a real run takes only when
hundreds of predecessors stay silent at once."""
import sys
sys.path.insert(0, '.')
import numpy as np, mods_amd
from oracle import pyoracle as O
rs = np.random.RandomState(4)
n1, n2 = 96, 70000
d2 = rs.randint(0, 120, (n2, 128)).astype(np.float32)
d1 = np.clip(d2[rs.choice(n2, n1)] + rs.randint(-3, 4, (n1, 128)), 0, 255).astype(np.float32)
pos2 = rs.uniform(0, 2000, (n2, 2))
ctx = mods_amd.Context(0)
for ratio in (0.8, 1.0):
    ref = O.match_fginn(d1, d2, pos2, ratio, 30.0, 50); got = ctx.match_fginn(d1, d2, pos2, ratio, 30.0, 50)
    ok = len(ref) == len(got) and all(np.array_equal(ref[f], got[f], equal_nan=True) if ref[f].dtype.kind == 'f' else np.array_equal(ref[f], got[f]) for f in ref.dtype.names)
    print("n2 = 70000, ratio", ratio, "IDENTICAL" if ok else "MISMATCH", len(ref))

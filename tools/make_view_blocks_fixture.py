#!/usr/bin/env python3
"""Records the per-view blocks (regions + u8 descriptors) the engine produces for a small image: the input of the
world-size-2 CPU test of the view-shard exchange (tests/test_distributed_cpu.py).  Run on the GPU box:
    gpurun -- 'python tools/make_view_blocks_fixture.py gpurun_out/view_blocks_small.npz'
and copy the file to tests/golden/."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mods_amd
from mods_amd import synthetic

out = sys.argv[1]
a, _, _ = synthetic.make_pair(rows=160, cols=208, nblobs=150, seed=4242)
ctx = mods_amd.Context(0)
views = mods_amd.set_vs_pars([1.0], [1, 2, 3, 4, 6], 360.0, 0.2, 1, [])
par = mods_amd.default_pair_params()
ia = ctx.upload(a)
regs, desc, counts = ctx.detect_describe_views(ia, views, par, want_counts=True)
# per-view blocks as the shards produce them: ids local to the block
blocks_regs, blocks_desc = [], []
for v in range(len(views)):
    r, d, c = ctx.detect_describe_views(ia, views, par, view_begin=v, view_step=len(views), want_counts=True)
    assert len(r) == counts[v]
    blocks_regs.append(r); blocks_desc.append(d.astype(np.uint8))
def raw(r):   # the 200-byte records exactly as the C ABI lays them out (np.save would drop the struct padding)
    r = np.ascontiguousarray(r, mods_amd.REGION)
    return np.frombuffer(r.tobytes(), np.uint8).reshape(len(r), mods_amd.REGION.itemsize).copy()
np.savez_compressed(out, regs=raw(regs), desc=desc.astype(np.uint8), counts=counts,
                    block_regs=np.concatenate([raw(r) for r in blocks_regs]), block_desc=np.concatenate(blocks_desc),
                    views=np.array([[v.zoom, v.tilt, v.phi] for v in views]))
print("views", len(views), "counts", counts.tolist(), "total", len(regs))

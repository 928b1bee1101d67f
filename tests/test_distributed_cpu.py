"""CPU, world_size 2 over gloo: the view-shard exchange on RECORDED engine output.

The native path (csrc/engine_shard.hip) does, per image side: all-gather of the per-view counts, all-gather of the padded
328-byte row blocks, then a gather into the reference's (view, detection) order given by modsx_view_block_order, then
AddRegionsToList's id re-basing.  RCCL needs GPUs, so here the same three steps run over gloo on the blocks the engine
produced for a small image (tests/golden/view_blocks_small.npz, written by tools/make_view_blocks_fixture.py on an MI355X):
the ordering function is the library's own (host code, no device needed), and the result must be the engine's unsharded
output, field for field.  Also covers the pair-shard bench plumbing (barrier, max-over-ranks time)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "view_blocks_small.npz")


def _blocks():
    import mods_amd
    z = np.load(FIX)
    counts = z["counts"].astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(counts)])
    R = mods_amd.REGION
    br = np.frombuffer(np.ascontiguousarray(z["block_regs"]).tobytes(), R)        # raw 200-byte C records
    blocks = [(br[starts[v]:starts[v + 1]], z["block_desc"][starts[v]:starts[v + 1]]) for v in range(len(counts))]
    ref = np.frombuffer(np.ascontiguousarray(z["regs"]).tobytes(), R)
    return blocks, counts, ref, z["desc"]


def _pack(regs, desc):
    import mods_amd
    n = len(regs)
    rows = np.zeros((n, mods_amd.REGION.itemsize + 128), np.uint8)
    if n:
        rows[:, :mods_amd.REGION.itemsize] = np.frombuffer(np.ascontiguousarray(regs).tobytes(), np.uint8).reshape(n, -1)
        rows[:, mods_amd.REGION.itemsize:] = desc
    return rows


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mods_amd
    from mods_amd import distributed as D
    blocks, counts_ref, ref_regs, ref_desc = _blocks()
    nviews = len(blocks)
    mine = D.shard_views(nviews, rank, world)
    rows_l = _pack(np.concatenate([blocks[v][0] for v in mine]), np.concatenate([blocks[v][1] for v in mine]))
    cnt = np.zeros(nviews, np.int32)
    for v in mine:
        cnt[v] = len(blocks[v][0])
    # 1. counts
    allc = [torch.zeros(nviews, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(allc, torch.from_numpy(cnt))
    counts = torch.stack(allc).numpy()
    # 2. padded row blocks
    src, maxrows = mods_amd.view_block_order(counts)
    pad = np.zeros((maxrows, rows_l.shape[1]), np.uint8)
    pad[:len(rows_l)] = rows_l
    parts = [torch.zeros((maxrows, rows_l.shape[1]), dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(pad))
    rows_all = torch.cat(parts).numpy()
    # 3. reference order + id re-basing by view counts
    rows = rows_all[src]
    R = mods_amd.REGION
    regs = np.frombuffer(np.ascontiguousarray(rows[:, :R.itemsize]).tobytes(), R, len(rows)).copy()
    desc = np.ascontiguousarray(rows[:, R.itemsize:])
    vc = np.array([counts[v % world, v] for v in range(nviews)])
    starts = np.concatenate([[0], np.cumsum(vc)[:-1]])
    for v in range(nviews):
        regs["id"][starts[v]:starts[v] + vc[v]] += starts[v]
        regs["parent_id"][starts[v]:starts[v] + vc[v]] += starts[v]
    ok = np.array_equal(vc, counts_ref) and len(regs) == len(ref_regs) and np.array_equal(desc, ref_desc)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import same_records
    ok = ok and same_records(regs, ref_regs)          # every field (struct padding bytes are not data)
    # bench plumbing: max-over-ranks of a per-rank time, sum of per-rank work
    t = torch.tensor([1.0 + rank, 10.0 * (rank + 1)], dtype=torch.float64)
    tmax, tsum = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    ok = ok and float(tmax[0]) == float(world) and float(tsum[1]) == 10.0 * world * (world + 1) / 2
    dist.barrier()
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_view_shard_exchange_on_recorded_blocks(world):
    port = 29500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_view_block_order_and_sharding():
    sys.path.insert(0, ROOT)
    import mods_amd
    from mods_amd import distributed as D
    assert D.shard_views(8, 1, 3) == [1, 4, 7]
    counts = np.array([[2, 0, 3, 0], [0, 1, 0, 0]])           # rank 0 owns views 0,2; rank 1 owns 1,3 (empty)
    idx, maxrows = mods_amd.view_block_order(counts)
    assert maxrows == 5 and idx.tolist() == [0, 1, 5, 2, 3, 4]
    # identity-like views all carry img_id 0: block boundaries must come from the counts, not from runs of img_id
    counts = np.array([[3, 0, 2], [0, 4, 0]])
    idx, maxrows = mods_amd.view_block_order(counts)
    assert maxrows == 5 and idx.tolist() == [0, 1, 2, 5, 6, 7, 8, 3, 4]

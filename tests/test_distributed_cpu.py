"""CPU, world_size 2 over gloo: the view-shard exchange (one all-gather of padded row blocks) rebuilds the
reference's list order, and the pair-shard bench plumbing (barrier, max-over-ranks time) works."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_blocks(nviews, seed=0):
    """Deterministic per-view blocks: counts vary per view, one view is empty."""
    import mods_amd
    rs = np.random.RandomState(seed)
    blocks = []
    for v in range(nviews):
        n = 0 if v == 3 else int(rs.randint(1, 40))
        regs = np.zeros(n, mods_amd.REGION)
        regs["img_id"] = v
        regs["det_kp"]["x"] = rs.uniform(0, 100, n)
        regs["reproj_kp"]["x"] = rs.uniform(0, 100, n)
        regs["reproj_kp"]["y"] = rs.uniform(0, 100, n)
        desc = rs.randint(0, 256, (n, 128)).astype(np.uint8)
        blocks.append((regs, desc))
    return blocks


def _worker(rank, world, port, nviews, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mods_amd import distributed as D
    blocks = _fake_blocks(nviews)
    mine = D.shard_views(nviews, rank, world)
    regs_l = np.concatenate([blocks[v][0] for v in mine])
    desc_l = np.concatenate([blocks[v][1] for v in mine])
    counts = np.zeros(nviews, np.int64)
    for v in mine:
        counts[v] = len(blocks[v][0])
    rows = torch.from_numpy(D.pack_rows(regs_l, desc_l))
    rows_g, allc = D.all_gather_view_blocks(rows, counts, nviews, torch.device("cpu"))
    regs_g, desc_g = D.unpack_rows(rows_g.numpy())
    regs_g = D.rebase_ids(regs_g)
    ref_regs = np.concatenate([b[0] for b in blocks])
    ref_desc = np.concatenate([b[1] for b in blocks])
    ok = (len(regs_g) == len(ref_regs) and np.array_equal(desc_g, ref_desc)
          and np.array_equal(regs_g["reproj_kp"]["x"], ref_regs["reproj_kp"]["x"])
          and np.array_equal(regs_g["img_id"], ref_regs["img_id"]))
    # ids re-based like AddRegionsToList: block start added to the (zero) local ids
    starts = np.concatenate([[0], np.cumsum([len(b[0]) for b in blocks])[:-1]])
    exp_ids = np.concatenate([np.full(len(b[0]), s) for b, s in zip(blocks, starts)])
    ok = ok and np.array_equal(regs_g["id"], exp_ids) and allc.shape == (world, nviews)
    # bench plumbing: max-over-ranks of a per-rank time, sum of per-rank work
    t = torch.tensor([1.0 + rank, 10.0 * (rank + 1)], dtype=torch.float64)
    tmax, tsum = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    ok = ok and float(tmax[0]) == float(world) and float(tsum[1]) == 10.0 * world * (world + 1) / 2
    dist.barrier()
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("nviews", [8, 11])
def test_view_shard_all_gather_world2(nviews):
    world = 2
    port = 29500 + (os.getpid() % 2000) + nviews
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, nviews, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_global_order_and_sharding():
    sys.path.insert(0, ROOT)
    from mods_amd import distributed as D
    assert D.shard_views(8, 1, 3) == [1, 4, 7]
    counts = np.array([[2, 0, 3, 0], [0, 1, 0, 0]])           # rank 0 owns views 0,2; rank 1 owns 1,3 (empty)
    idx, maxrows = D.global_order(counts, 2)
    assert maxrows == 5 and idx.tolist() == [0, 1, 5, 2, 3, 4]

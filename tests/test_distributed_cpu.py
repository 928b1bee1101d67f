"""CPU, world sizes 2 and 3 over gloo: the view-shard exchange in the SHIPPED block format, on recorded engine output.

One exchange of the native path (csrc/engine_shard.hip) = every rank packs a block (header {magic, rc, rows, items, per-item
counts} + rows of region + descriptors, padded to the agreed block size), ONE all-gather of the blocks, then the reference's
(image, view, detection) order is rebuilt from the gathered headers.  RCCL needs GPUs; here the ranks are gloo processes and the
two data steps are the library's own host statements of the wire format -- modsx_shard_block_pack / modsx_shard_blocks_unpack,
the functions the GPU tests hold the device kernels (k_pack_rows, k_unpack_blocks) against byte for byte.  Input: the per-view
blocks the engine produced for a small image (tests/golden/view_blocks_small.npz, tools/make_view_blocks_fixture.py on an
MI355X).  Covered: one image and a two-image item list (the batched pair call), one and two descriptor classes per row, the
block-retry protocol (a first block size that is too small is seen by every rank in the headers; all grow alike and repeat), a
failing rank (its rc travels in the header and every rank gets it), AddRegionsToList's id re-basing, and the bench plumbing
(barrier, max-over-ranks time)."""
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


def _exchange(mods_amd, rank, world, item_blocks, ndesc, first_rows, fail_rank=-1, fmt=0):
    """One exchange as engine_shard.hip runs it: pack, all-gather, unpack; a block that was too small is repeated with the size every
    rank derives from the gathered headers.  item_blocks[f] = (regs, desc) of item f.  Returns (regs, descs, counts, retries)."""
    items = len(item_blocks)
    mine = [f for f in range(items) if f % world == rank]
    R = mods_amd.REGION
    regs_l = np.concatenate([item_blocks[f][0] for f in mine]) if mine else np.zeros(0, R)
    d0 = np.concatenate([item_blocks[f][1] for f in mine]) if mine else np.zeros((0, 128), np.uint8)
    descs_l = [d0] + [np.ascontiguousarray(255 - d0) for _ in range(ndesc - 1)]      # class k > 0: a second, different descriptor
    cnt = np.zeros(items, np.int32)
    for f in mine:
        cnt[f] = len(item_blocks[f][0])
    rows, retries = first_rows, 0
    while True:
        blk = mods_amd.shard_block_pack(regs_l, descs_l, cnt, rows, rc_local=(-3 if rank == fail_rank else 0), row_format=fmt)
        parts = [torch.zeros(len(blk), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(blk))
        allb = torch.cat(parts).numpy()
        regs, descs, got = mods_amd.shard_blocks_unpack(allb, world, items, rows, ndesc, row_format=fmt)
        if regs is None:                 # every rank sees the overflow in the same headers and takes the same new size
            rows = got + got // 4 + 64
            retries += 1
            assert retries < 3
            continue
        return regs, descs, got, retries


def _rebase(regs, counts):
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    for v in range(len(counts)):
        regs["id"][starts[v]:starts[v] + counts[v]] += starts[v]
        regs["parent_id"][starts[v]:starts[v] + counts[v]] += starts[v]
    return regs


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mods_amd
    from common import same_records
    blocks, counts_ref, ref_regs, ref_desc = _blocks()
    nv = len(blocks)
    ok = True
    # 1. one image, one descriptor class, a first block size below every rank's row count: one retry, then the engine's list
    regs, descs, cnt, retries = _exchange(mods_amd, rank, world, blocks, 1, first_rows=7)
    ok = ok and retries == 1 and np.array_equal(cnt, counts_ref) and np.array_equal(descs[0], ref_desc)
    ok = ok and same_records(_rebase(regs.copy(), cnt), ref_regs)
    # 2. two images in one exchange (the batched pair call: items = image * nviews + view), two descriptor classes per row
    two = blocks + blocks[::-1]
    regs2, descs2, cnt2, retries2 = _exchange(mods_amd, rank, world, two, 2, first_rows=4096)
    exp_regs = np.concatenate([b[0] for b in two]); exp_desc = np.concatenate([b[1] for b in two])
    ok = ok and retries2 == 0 and np.array_equal(cnt2, [len(b[0]) for b in two])
    ok = ok and same_records(regs2, exp_regs) and np.array_equal(descs2[0], exp_desc) and np.array_equal(descs2[1], 255 - exp_desc)
    ok = ok and same_records(_rebase(regs2[:len(ref_regs)].copy(), cnt2[:nv]), ref_regs)        # image 0 of the item list
    # 2b. the same exchange in the pair call's row format (56 B of geometry + descriptors): what modsx_match_pairs_views_sharded moves
    kp2, dk2, ck2, rk2 = _exchange(mods_amd, rank, world, two, 2, first_rows=9, fmt=mods_amd.SHARD_ROW_KP)
    ok = ok and rk2 == 1 and np.array_equal(ck2, cnt2) and np.array_equal(kp2, mods_amd.shard_kp_rows(exp_regs))
    ok = ok and np.array_equal(dk2[0], exp_desc) and np.array_equal(dk2[1], 255 - exp_desc)
    # 3. a failing rank: its rc travels in the header, every rank raises the same error from the same call
    try:
        _exchange(mods_amd, rank, world, blocks, 1, first_rows=4096, fail_rank=world - 1)
        ok = False
    except RuntimeError as e:
        ok = ok and "(-3)" in str(e)
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
def test_view_shard_exchange_in_the_shipped_block_format(world):
    port = 29500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _owner_worker(rank, world, port, ret):
    """The owner-only exchange (modsx_comm_set_exchange(MODSX_EXCHANGE_OWNER)) over gloo: counts to every rank, then the messages
    modsx_shard_owner_plan states -- rows of image j to owner[j] alone -- as point-to-point sends and receives, and the plan's jobs put
    the received rows at the list positions the all-gather path gives them."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mods_amd
    blocks, counts_ref, ref_regs, ref_desc = _blocks()
    nviews = len(blocks)
    item_blocks = blocks + blocks[::-1] + blocks            # three images
    owners = [0, world - 1, 1 % world]
    items = len(item_blocks)
    KP = mods_amd.SHARD_ROW_KP
    row_b = 56 + 128
    mine = [f for f in range(items) if f % world == rank]
    regs_l = np.concatenate([item_blocks[f][0] for f in mine])
    desc_l = np.concatenate([item_blocks[f][1] for f in mine])
    cnt = np.zeros(items, np.int32)
    for f in mine:
        cnt[f] = len(item_blocks[f][0])
    hdr_b = mods_amd.shard_block_bytes(items, 0, 1, KP)
    blk = mods_amd.shard_block_pack(regs_l, [desc_l], cnt, len(regs_l), row_format=KP)      # the rows a rank packs, in item order
    rows_l = blk[hdr_b:].reshape(len(regs_l), row_b)
    # counts to every rank (the device path all-gathers the headers; the counts are all it reads from them)
    parts = [torch.zeros(items, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(cnt))
    counts = torch.stack(parts).sum(0).numpy().astype(np.int32)
    ok = np.array_equal(counts, [len(b[0]) for b in item_blocks])
    plan = mods_amd.shard_owner_plan(counts, nviews, world, rank, owners)
    recv = np.zeros((max(1, plan["recv_rows"]), row_b), np.uint8)
    # gloo has no send to self: a rank's own messages are copied, in order, like the k-th send meets the k-th receive of a peer
    self_sends = [m for m in plan["sends"] if m[0] == rank]
    self_recvs = [m for m in plan["recvs"] if m[0] == rank]
    ok = ok and len(self_sends) == len(self_recvs)
    for sm, rm in zip(self_sends, self_recvs):
        ok = ok and sm[3] == rm[3] and sm[1] == rm[1]
        recv[rm[2]:rm[2] + rm[3]] = rows_l[sm[2]:sm[2] + sm[3]]
    reqs, bufs = [], []
    for peer, image, row0, rows in plan["recvs"]:
        if peer != rank:
            t = torch.zeros(rows * row_b, dtype=torch.uint8)
            bufs.append((t, row0, rows))
            reqs.append(dist.irecv(t, src=int(peer)))
    for peer, image, row0, rows in plan["sends"]:
        if peer != rank:
            ok = ok and owners[image] == peer
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(rows_l[row0:row0 + rows]).reshape(-1)), dst=int(peer)))
    for q in reqs:
        q.wait()
    for t, row0, rows in bufs:
        recv[row0:row0 + rows] = t.numpy().reshape(rows, row_b)
    lst = np.zeros((plan["list_rows"], row_b), np.uint8)
    filled = np.zeros(plan["list_rows"], bool)
    for src0, dst0, n, _ in plan["jobs"]:
        lst[dst0:dst0 + n] = recv[src0:src0 + n]
        filled[dst0:dst0 + n] = True
    # an owner holds exactly its images, at the positions of the full list, with the rows the all-gather path would give it
    exp_regs = np.concatenate([b[0] for b in item_blocks]); exp_desc = np.concatenate([b[1] for b in item_blocks])
    exp_kp = mods_amd.shard_kp_rows(exp_regs)
    start = np.concatenate([[0], np.cumsum(counts)])
    for j in range(3):
        lo, hi = start[j * nviews], start[(j + 1) * nviews]
        if owners[j] == rank:
            ok = ok and filled[lo:hi].all()
            ok = ok and np.array_equal(lst[lo:hi, :56].copy().view(np.float64).reshape(-1, 7), exp_kp[lo:hi])
            ok = ok and np.array_equal(lst[lo:hi, 56:], exp_desc[lo:hi])
        else:
            ok = ok and not filled[lo:hi].any()
    ok = ok and plan["recv_rows"] == sum(int(start[(j + 1) * nviews] - start[j * nviews]) for j in range(3) if owners[j] == rank)
    dist.barrier()
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_owner_only_exchange_over_gloo(world):
    port = 31500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_owner_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_owner_plan_single_process():
    """modsx_shard_owner_plan by hand on a small case: 2 images x 3 views over 2 ranks, image 0 read by rank 1, image 1 by rank 0."""
    sys.path.insert(0, ROOT)
    import mods_amd
    counts = [2, 1, 3, 0, 4, 5]          # items 0..5; rank 0 holds items 0, 2, 4 (local rows 2 + 3 + 4), rank 1 items 1, 3, 5 (1 + 0 + 5)
    p0 = mods_amd.shard_owner_plan(counts, 3, 2, 0, [1, 0])
    p1 = mods_amd.shard_owner_plan(counts, 3, 2, 1, [1, 0])
    assert p0["sends"].tolist() == [[1, 0, 0, 5], [0, 1, 5, 4]]          # image 0: local rows 0..4 to rank 1; image 1: rows 5..8 to itself
    assert p1["sends"].tolist() == [[1, 0, 0, 1], [0, 1, 1, 5]]
    assert p0["recvs"].tolist() == [[0, 1, 0, 4], [1, 1, 4, 5]] and p0["recv_rows"] == 9      # image 1: from rank 0, then from rank 1
    assert p1["recvs"].tolist() == [[0, 0, 0, 5], [1, 0, 5, 1]] and p1["recv_rows"] == 6
    assert p0["list_rows"] == p1["list_rows"] == 15
    # jobs: items 3 (empty), 4, 5 of image 1 on rank 0; items 0, 1, 2 of image 0 on rank 1 (list rows count all items)
    assert p0["jobs"].tolist() == [[0, 6, 4, 0], [4, 10, 5, 0]]
    assert p1["jobs"].tolist() == [[0, 0, 2, 0], [5, 2, 1, 0], [2, 3, 3, 0]]


def test_block_format_single_process():
    """The host statement of the wire format without any transport: sizes, the header, a too-small block, capacity, and that the
    ordering is the one modsx_view_block_order states."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mods_amd
    from common import same_records
    blocks, counts_ref, ref_regs, ref_desc = _blocks()
    nv, world = len(blocks), 3
    assert mods_amd.shard_block_bytes(nv, 10, 1) == ((4 + nv) * 4 + 63) // 64 * 64 + 10 * 328
    assert mods_amd.shard_block_bytes(nv, 10, 2) == ((4 + nv) * 4 + 63) // 64 * 64 + 10 * 456
    allb, counts = [], np.zeros((world, nv), np.int32)
    rows = int(max(sum(len(blocks[v][0]) for v in range(r, nv, world)) for r in range(world)))
    for r in range(world):
        mine = list(range(r, nv, world))
        cnt = np.zeros(nv, np.int32)
        for v in mine:
            cnt[v] = counts[r, v] = len(blocks[v][0])
        blk = mods_amd.shard_block_pack(np.concatenate([blocks[v][0] for v in mine]), [np.concatenate([blocks[v][1] for v in mine])], cnt, rows)
        hdr = np.frombuffer(blk[:16].tobytes(), np.int32)
        assert hdr[0] == 0x4D585348 and hdr[1] == 0 and hdr[2] == cnt.sum() and hdr[3] == nv
        allb.append(blk)
    regs, descs, cnt = mods_amd.shard_blocks_unpack(np.concatenate(allb), world, nv, rows, 1)
    assert np.array_equal(cnt, counts_ref) and np.array_equal(descs[0], ref_desc)
    # the same positions modsx_view_block_order names
    src, maxrows = mods_amd.view_block_order(counts)
    assert maxrows == rows
    hdrB = mods_amd.shard_block_bytes(nv, 0, 1)
    rows_all = np.concatenate([b[hdrB:].reshape(rows, 328) for b in allb])
    by_order = np.frombuffer(np.ascontiguousarray(rows_all[src][:, :200]).tobytes(), mods_amd.REGION)
    assert same_records(regs, by_order)          # field-wise (struct padding bytes are not data)
    # too small a block: the header keeps the true count, the unpack reports what is needed
    small = [mods_amd.shard_block_pack(np.concatenate([blocks[v][0] for v in range(r, nv, world)]),
                                       [np.concatenate([blocks[v][1] for v in range(r, nv, world)])], counts[r], 5) for r in range(world)]
    r_, d_, need = mods_amd.shard_blocks_unpack(np.concatenate(small), world, nv, 5, 1)
    assert r_ is None and need == rows
    # the pair call's row format: 56 bytes of geometry (x, y, a11, a12, a21, a22, s of reproj_kp) instead of the whole region
    KP = mods_amd.SHARD_ROW_KP
    assert mods_amd.shard_block_bytes(nv, 10, 1, KP) == ((4 + nv) * 4 + 63) // 64 * 64 + 10 * 184
    assert mods_amd.shard_block_bytes(nv, 10, 2, KP) == ((4 + nv) * 4 + 63) // 64 * 64 + 10 * 312
    allk = [mods_amd.shard_block_pack(np.concatenate([blocks[v][0] for v in range(r, nv, world)]),
                                      [np.concatenate([blocks[v][1] for v in range(r, nv, world)])], counts[r], rows, row_format=KP)
            for r in range(world)]
    kp, dk, ck = mods_amd.shard_blocks_unpack(np.concatenate(allk), world, nv, rows, 1, row_format=KP)
    assert np.array_equal(ck, counts_ref) and np.array_equal(dk[0], ref_desc)
    assert kp.shape == (len(regs), 7) and np.array_equal(kp, mods_amd.shard_kp_rows(regs))


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

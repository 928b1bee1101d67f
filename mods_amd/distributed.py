"""View sharding across the GPUs of a node.

Each (image, view) is independent from view synthesis through description
(imagerepresentation.cpp:622-2043); the only exchange is the ordered concatenation of the per-view
blocks (:2044-2045).  Rank r takes the views v with v % world == r, runs them through libmodsx on its
own GPU, and ONE all-gather per image side moves region records + u8 descriptors (328 B per region,
padded to the longest shard) over RCCL/xGMI.  Every rank then rebuilds the reference's list order
(view index, detection order), matching shards the query rows, and rank 0 verifies.

The collective code is backend agnostic (torch.distributed): `nccl` (= RCCL) with device tensors on the
GPU box, `gloo` with CPU tensors in the world_size-2 CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import REGION

ROW_BYTES = REGION.itemsize + 128   # one gathered row: the region record followed by its descriptor


def shard_views(nviews, rank, world):
    return list(range(rank, nviews, world))


def pack_rows(regs, desc_u8):
    """[n, 328] uint8 rows = REGION bytes | 128 descriptor bytes."""
    n = len(regs)
    rows = np.zeros((n, ROW_BYTES), np.uint8)
    if n:
        rows[:, :REGION.itemsize] = np.frombuffer(np.ascontiguousarray(regs, REGION).tobytes(), np.uint8).reshape(n, -1)
        rows[:, REGION.itemsize:] = desc_u8
    return rows


def unpack_rows(rows):
    n = len(rows)
    regs = np.frombuffer(np.ascontiguousarray(rows[:, :REGION.itemsize]).tobytes(), REGION, n).copy() if n else np.zeros(0, REGION)
    desc = np.ascontiguousarray(rows[:, REGION.itemsize:]) if n else np.zeros((0, 128), np.uint8)
    return regs, desc


def global_order(counts_per_rank_view, world):
    """Index list that reorders the concatenation of the (padded) rank blocks into view order.

    counts_per_rank_view: [world, nviews] (zero for views a rank does not own).  Returns (index, offsets)
    where index[i] = (rank, row-in-rank-block) flattened as rank * maxrows + row."""
    counts = np.asarray(counts_per_rank_view)
    nviews = counts.shape[1]
    totals = counts.sum(1)
    maxrows = int(totals.max()) if len(totals) else 0
    starts = np.zeros_like(counts)
    for r in range(world):
        starts[r] = np.concatenate([[0], np.cumsum(counts[r])[:-1]])
    idx = []
    for v in range(nviews):
        r = v % world
        c = int(counts[r, v])
        if c:
            idx.append(r * maxrows + int(starts[r, v]) + np.arange(c))
    return (np.concatenate(idx) if idx else np.zeros(0, np.int64)), maxrows


def rebase_ids(regs):
    """AddRegionsToList (imagerepresentation.cpp:588-600): ids of each view block += size of the list so far."""
    regs = regs.copy()
    start = 0
    n = len(regs)
    while start < n:
        end = start
        while end < n and regs["img_id"][end] == regs["img_id"][start]:
            end += 1
        regs["id"][start:end] += start
        regs["parent_id"][start:end] += start
        start = end
    return regs


def all_gather_view_blocks(rows_local, counts_local, nviews, device, group=None):
    """One all-gather of the padded row blocks (+ one tiny all-gather of the per-view counts).

    rows_local: [n_local, 328] uint8 torch tensor on `device` (view blocks of this rank, in view order);
    counts_local: [nviews] int64 (regions per owned view).  Returns (rows_global [N,328] in reference order,
    counts [world, nviews])."""
    world = dist.get_world_size(group)
    cnt = torch.as_tensor(np.asarray(counts_local, np.int64), device=device)
    all_cnt = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(all_cnt, cnt, group=group)
    counts = torch.stack(all_cnt).cpu().numpy()
    index, maxrows = global_order(counts, world)
    pad = torch.zeros((maxrows, ROW_BYTES), dtype=torch.uint8, device=device)
    if len(rows_local):
        pad[: len(rows_local)] = rows_local
    gathered = torch.empty((world * maxrows, ROW_BYTES), dtype=torch.uint8, device=device)
    if maxrows:
        dist.all_gather_into_tensor(gathered, pad, group=group) if hasattr(dist, "all_gather_into_tensor") and device.type != "cpu" \
            else _all_gather_list(gathered, pad, world, maxrows, group)
    idx = torch.as_tensor(index, device=device, dtype=torch.long)
    return gathered.index_select(0, idx), counts


def _all_gather_list(out, pad, world, maxrows, group):
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    for r in range(world):
        out[r * maxrows:(r + 1) * maxrows] = parts[r]


def detect_describe_views_sharded(ctx, img, views, params, group=None):
    """Per-rank view shard on the local GPU + the all-gather.  Returns (regions in reference order with re-based
    ids, u8 descriptor tensor [N,128] on the GPU)."""
    import mods_amd
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = torch.device("cuda", torch.cuda.current_device())
    cap = 1 << 16
    while True:
        desc_dev = torch.empty((cap, 128), dtype=torch.uint8, device=device)
        try:
            regs, _, counts = ctx.detect_describe_views(img, views, params, view_begin=rank, view_step=world,
                                                        want_desc=False, dev_desc=desc_dev.data_ptr(), dev_cap=cap,
                                                        want_counts=True)
            break
        except RuntimeError as e:
            if "too small" in str(e) and cap < (1 << 24):
                cap *= 4
                continue
            raise
    n = len(regs)
    reg_bytes = torch.from_numpy(np.frombuffer(np.ascontiguousarray(regs, REGION).tobytes(), np.uint8).reshape(n, -1).copy()) \
        if n else torch.zeros((0, REGION.itemsize), dtype=torch.uint8)
    rows = torch.empty((n, ROW_BYTES), dtype=torch.uint8, device=device)
    if n:
        rows[:, :REGION.itemsize] = reg_bytes.to(device)
        rows[:, REGION.itemsize:] = desc_dev[:n]
    rows_g, _ = all_gather_view_blocks(rows, counts, len(views), device, group)
    regs_g = np.frombuffer(rows_g[:, :REGION.itemsize].contiguous().cpu().numpy().tobytes(), REGION, len(rows_g)).copy() \
        if len(rows_g) else np.zeros(0, REGION)
    desc_g = rows_g[:, REGION.itemsize:].contiguous()
    # with world == 1 the library call covered all views and already re-based the ids (modsx.h)
    return (regs_g if world == 1 else rebase_ids(regs_g)), desc_g


def match_sharded(ctx, regs1, desc1, regs2, desc2, ratio, contrad_dist, nn=50, group=None):
    """Query rows of image 1 are split over the ranks; every rank holds all of image 2.  Tentatives are
    gathered on every rank in query order (the reference's order)."""
    import mods_amd
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n1 = len(regs1)
    lo, hi = (n1 * rank) // world, (n1 * (rank + 1)) // world
    pos2 = np.stack([regs2["reproj_kp"]["x"], regs2["reproj_kp"]["y"]], 1) if len(regs2) else np.zeros((0, 2))
    if hi > lo and len(regs2):
        t = ctx.match_fginn_device(desc1.data_ptr() + lo * 128, hi - lo, desc2.data_ptr(), len(regs2), pos2, ratio,
                                   contrad_dist, nn)
        t["q"] += lo
    else:
        t = np.zeros(0, mods_amd.TENT)
    parts = [None] * world
    dist.all_gather_object(parts, t, group=group)
    return np.concatenate(parts)

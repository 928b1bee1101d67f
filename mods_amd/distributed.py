"""View sharding across the GPUs of a node: a thin caller of the native path (csrc/engine_shard.hip).

Each (image, view) is independent from view synthesis through description (imagerepresentation.cpp:612-622); the only
exchange is the ordered concatenation of the per-view blocks (:2044-2045).  Rank r takes the views v with v % world == r;
libmodsx all-gathers the region rows + u8 descriptors (328 B per region, behind a header with the per-view counts) and the
matcher's per-query result rows with ncclAllGather on device buffers (RCCL over xGMI).  ONE communicator per rank: the
contexts of a rank are its lanes and the library issues their collectives in round-robin lane order, the same on every
rank.  Nothing here touches the data: this module bootstraps the communicator (the 128-byte id travels over
torch.distributed, or not at all when world == 1), and forwards the calls.
"""
import ctypes as C
import threading

import mods_amd


def shard_views(nviews, rank, world):
    return list(range(rank, nviews, world))


class NativeComm:
    """The communicator of this rank; context w of `ctxs` is lane w.  All ranks must create the same number of contexts
    and drive lane w with the same sequence of calls, every lane the same number of calls per round."""

    def __init__(self, ctxs, dist=None, rank=None, world=None, uid=None, timeout_ms=None):
        self.ctxs = list(ctxs)
        self.dist = dist
        self.rank = (dist.get_rank() if dist is not None else 0) if rank is None else rank
        self.world = (dist.get_world_size() if dist is not None else 1) if world is None else world
        if uid is None:
            box = [mods_amd.comm_unique_id() if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(box, src=0)      # control plane only: 128 bytes
            uid = box[0]
        self.comm = self.ctxs[0].comm_create(uid, self.rank, self.world)
        L = mods_amd.lib()
        if len(self.ctxs) > 1:
            mods_amd._check(L.modsx_comm_set_lanes(C.c_void_p(self.comm), len(self.ctxs)), "comm_set_lanes")
        for w, c in enumerate(self.ctxs):
            mods_amd._check(L.modsx_comm_attach(C.c_void_p(self.comm), c._c(), w), "comm_attach")
        if timeout_ms:
            mods_amd._check(L.modsx_comm_set_timeout(C.c_void_p(self.comm), int(timeout_ms)), "comm_set_timeout")

    def detect_describe_views_sharded(self, w, img, views, params):
        return self.ctxs[w].detect_describe_views_sharded(self.comm, img, views, params)

    def match_fginn_sharded(self, w, d1_ptr, n1, d2_ptr, n2, pos2, ratio=0.8, contrad_dist=30.0, nn=50):
        return self.ctxs[w].match_fginn_sharded(self.comm, d1_ptr, n1, d2_ptr, n2, pos2, ratio, contrad_dist, nn)

    def match_pair_views_sharded(self, w, img1, img2, views, params, owner=0):
        """Every rank calls this; the result of the owner rank carries the verified correspondences, the others get None."""
        r = self.ctxs[w].match_pair_views_sharded(self.comm, img1, img2, views, params, owner)
        return r if owner < 0 or owner == self.rank else None

    def match_pairs_views_sharded(self, w, imgs1, imgs2, views, params, owner_base=0, arrays=True):
        """Up to 16 pairs in one sharded call on lane w: one exchange for all image sides; pair g is matched and verified by rank
        (owner_base + g) % world (non-owners: region counts only); owner_base < 0: every rank returns every pair (row-split
        matching, one result all-gather per descriptor class)."""
        return self.ctxs[w].match_pairs_views_sharded(self.comm, imgs1, imgs2, views, params, owner_base, arrays)

    def match_ladder_sharded(self, w, img1, img2, steps, params, min_matches=10):
        return self.ctxs[w].match_ladder(img1, img2, steps, params, min_matches, comm=self.comm)

    def set_exchange(self, mode):
        """mods_amd.EXCHANGE_OWNER: the rows of a pair travel to its owner rank only (modsx_comm_set_exchange; alike on every rank)."""
        mods_amd._check(mods_amd.lib().modsx_comm_set_exchange(C.c_void_p(self.comm), int(mode)), "comm_set_exchange")

    def lane_done(self, w):
        mods_amd.lib().modsx_comm_lane_done(C.c_void_p(self.comm), int(w))

    def reset_lanes(self):
        mods_amd.lib().modsx_comm_reset_lanes(C.c_void_p(self.comm))

    def describe(self):
        rk, wd, ver, byts, ncol = C.c_int(), C.c_int(), C.c_int(), C.c_long(), C.c_long()
        mods_amd.lib().modsx_comm_info(C.c_void_p(self.comm), C.byref(rk), C.byref(wd), C.byref(ver), C.byref(byts), C.byref(ncol))
        st = mods_amd.comm_stats(self.comm)
        return {"ranks_seen_by_rccl": wd.value, "rccl_version": ver.value, "communicators_per_rank": 1, "lanes": len(self.ctxs),
                "all_gather_calls_rank0": ncol.value, "bytes_all_gathered_rank0": byts.value,
                "block_retries": st["block_retries"], "agreement_collectives": st["agreements"],
                "owner_exchanges_rank0": st["exchanges"], "bytes_received_owner_exchange_rank0": st["bytes_received"],
                "transport": "loopback" if st["loopback"] else "rccl"}

    def close(self):
        if self.comm:
            mods_amd.lib().modsx_comm_destroy(C.c_void_p(self.comm))
        self.comm = None


def run_loopback(world, fn, lanes=1, device=0, timeout_ms=None):
    """Run `fn(rank, comm: NativeComm)` on `world` in-process ranks (one thread + `lanes` contexts each) over the loopback
    transport; returns the list of results by rank.  An exception on any rank is re-raised after all threads ended."""
    uid = mods_amd.comm_loopback_id(world)
    ctxs = [[mods_amd.Context(device) for _ in range(lanes)] for _ in range(world)]
    comms = [None] * world
    for r in range(world):
        comms[r] = NativeComm(ctxs[r], rank=r, world=world, uid=uid, timeout_ms=timeout_ms)
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r, comms[r])
        except BaseException as e:   # noqa: BLE001 -- reported to the caller below
            err[r] = e

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r in range(world):
        comms[r].close()
        for c in ctxs[r]:
            c.close()
    for e in err:
        if e is not None:
            raise e
    return out


def run_mock_rccl(world, fn, lanes=1, device=0, timeout_ms=None):
    """run_loopback's twin for the RCCL branch: the ranks are made with an id from `modsx_comm_unique_id` while an in-process stand-in
    fills librccl's function table (`modsx_debug_mock_rccl`), so `modsx_comm_create` and every collective take the code path a real
    multi-GPU run takes -- everything but librccl itself.  Tests only."""
    L = mods_amd.lib()
    mods_amd._check(L.modsx_debug_mock_rccl(1), "debug_mock_rccl")
    try:
        uid = mods_amd.comm_unique_id()
        ctxs = [[mods_amd.Context(device) for _ in range(lanes)] for _ in range(world)]
        comms = [NativeComm(ctxs[r], rank=r, world=world, uid=uid, timeout_ms=timeout_ms) for r in range(world)]
        out, err = [None] * world, [None] * world

        def body(r):
            try:
                out[r] = fn(r, comms[r])
            except BaseException as e:   # noqa: BLE001 -- reported to the caller below
                err[r] = e

        th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for r in range(world):
            comms[r].close()
            for c in ctxs[r]:
                c.close()
    finally:
        L.modsx_debug_mock_rccl(0)
    for e in err:
        if e is not None:
            raise e
    return out

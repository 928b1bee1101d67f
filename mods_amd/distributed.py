"""View sharding across the GPUs of a node: a thin caller of the native path (csrc/engine_shard.hip).

Each (image, view) is independent from view synthesis through description (imagerepresentation.cpp:612-622); the only
exchange is the ordered concatenation of the per-view blocks (:2044-2045).  Rank r takes the views v with v % world == r;
libmodsx all-gathers the region rows + u8 descriptors (328 B per region) and the matcher's per-query result rows with
ncclAllGather on device buffers (RCCL over xGMI), one communicator per context / stream.  Nothing here touches the data:
this module only bootstraps the communicators (the 128-byte RCCL id travels over torch.distributed, or not at all when
world == 1) and forwards the calls.
"""
import numpy as np

import mods_amd


def shard_views(nviews, rank, world):
    return list(range(rank, nviews, world))


class NativeComm:
    """One RCCL communicator per context (a context = one host thread + one HIP stream).  All ranks must create the
    same number of contexts and drive context w with the same sequence of calls."""

    def __init__(self, ctxs, dist=None, rank=None, world=None):
        self.ctxs = list(ctxs)
        self.dist = dist
        self.rank = (dist.get_rank() if dist is not None else 0) if rank is None else rank
        self.world = (dist.get_world_size() if dist is not None else 1) if world is None else world
        self.comms = []
        for c in self.ctxs:
            uid = [mods_amd.comm_unique_id() if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(uid, src=0)      # control plane only: 128 bytes per communicator
            self.comms.append(c.comm_create(uid[0], self.rank, self.world))

    def detect_describe_views_sharded(self, w, img, views, params):
        return self.ctxs[w].detect_describe_views_sharded(self.comms[w], img, views, params)

    def match_fginn_sharded(self, w, d1_ptr, n1, d2_ptr, n2, pos2, ratio=0.8, contrad_dist=30.0, nn=50):
        return self.ctxs[w].match_fginn_sharded(self.comms[w], d1_ptr, n1, d2_ptr, n2, pos2, ratio, contrad_dist, nn)

    def match_pair_views_sharded(self, w, img1, img2, views, params, owner=0):
        """Every rank calls this; the result of the owner rank carries the verified correspondences, the others get None."""
        r = self.ctxs[w].match_pair_views_sharded(self.comms[w], img1, img2, views, params, owner)
        return r if owner < 0 or owner == self.rank else None

    def describe(self):
        import ctypes as C
        rk, wd, ver, byts, ncol = C.c_int(), C.c_int(), C.c_int(), C.c_long(), C.c_long()
        tot_b = tot_c = 0
        for h in self.comms:
            mods_amd.lib().modsx_comm_info(C.c_void_p(h), C.byref(rk), C.byref(wd), C.byref(ver), C.byref(byts), C.byref(ncol))
            tot_b += byts.value
            tot_c += ncol.value
        return {"ranks_seen_by_rccl": wd.value, "rccl_version": ver.value, "communicators_per_rank": len(self.comms),
                "all_gather_calls_rank0": tot_c, "bytes_all_gathered_rank0": tot_b}

    def close(self):
        import ctypes as C
        for h in self.comms:
            mods_amd.lib().modsx_comm_destroy(C.c_void_p(h))
        self.comms = []

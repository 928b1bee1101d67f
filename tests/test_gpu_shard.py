"""GPU: the native view-sharded path (csrc/engine_shard.hip) at world sizes > 1 on ONE device.

The loopback transport runs W ranks inside this process (one host thread + context per rank); its all-gather is W
device-to-device copies, everything above it -- blocks with headers, device-side reference order, id re-basing, the
query-row split of the match, lanes, error agreement, watchdog -- is the code the RCCL transport runs.  Every result is
compared with the unsharded library calls (which tests/test_gpu_views.py compares with the oracle)."""
import os
import time

import numpy as np
import pytest

from common import dev_to_host, same_records

pytestmark = pytest.mark.gpu


def _views(modsx):
    return modsx.set_vs_pars([1.0], [1, 2, 3, 4, 6], 360.0, 0.2, 1, [])      # 8 views


def _same_pair_result(got, ref):
    assert got["n_regions"] == ref["n_regions"] and got["n_tentatives"] == ref["n_tentatives"]
    assert got["n_unique"] == ref["n_unique"] and got["n_verified"] == ref["n_verified"]
    assert got["ransac_samples"] == ref["ransac_samples"]
    assert same_records(got["tentatives"], ref["tentatives"])
    assert np.array_equal(got["ransac_inlier"], ref["ransac_inlier"]) and np.array_equal(got["verified"], ref["verified"])
    assert np.array_equal(got["H"], ref["H"])


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_equals_unsharded(ctx, modsx, small_pair, world):
    """Regions (every field, re-based ids), per-view counts, the HBM-resident descriptors, the tentatives of the sharded
    matcher and the whole pair result on every rank == the unsharded calls.  World 8 has as many ranks as views (one view per
    rank, the identity view alone on rank 0); world 3 leaves ranks with unequal view counts and a ragged last query block."""
    from mods_amd import distributed as D
    a, b, _ = small_pair
    views = _views(modsx)
    par = modsx.default_pair_params(ransac_seed=4)
    ia, ib = ctx.upload(a), ctx.upload(b)
    ref1, refd1, c1 = ctx.detect_describe_views(ia, views, par, want_counts=True)
    ref2, refd2 = ctx.detect_describe_views(ib, views, par)
    pos2 = np.stack([ref2["reproj_kp"]["x"], ref2["reproj_kp"]["y"]], 1)
    reft = ctx.match_fginn(refd1, refd2, pos2, 0.8, 30.0)
    ref = ctx.match_pair_views(ia, ib, views, par)
    import torch
    d2 = torch.from_numpy(refd2.astype(np.uint8)).cuda()

    def rank_body(r, comm):
        r1, d1ptr, cnt = comm.detect_describe_views_sharded(0, ia, views, par)
        desc = dev_to_host(d1ptr, len(r1) * 128).reshape(-1, 128)
        tent = comm.match_fginn_sharded(0, d1ptr, len(r1), d2.data_ptr(), len(ref2), pos2, 0.8, 30.0)
        pair_all = comm.ctxs[0].match_pair_views_sharded(comm.comm, ia, ib, views, par, -1)     # every rank verifies
        pair_own = comm.ctxs[0].match_pair_views_sharded(comm.comm, ia, ib, views, par, world - 1)
        return r1, cnt, desc, tent, pair_all, pair_own, comm.describe()

    out = D.run_loopback(world, rank_body)
    for r, (r1, cnt, desc, tent, pair_all, pair_own, info) in enumerate(out):
        assert same_records(r1, ref1), r
        assert np.array_equal(cnt, c1)
        assert np.array_equal(desc, refd1.astype(np.uint8)), r
        assert same_records(tent, reft), r
        _same_pair_result(pair_all, ref)
        if r == world - 1:
            _same_pair_result(pair_own, ref)
        else:
            assert pair_own["n_regions"] == ref["n_regions"] and pair_own["n_tentatives"] == ref["n_tentatives"]
            assert pair_own["n_verified"] == 0
        assert info["transport"] == "loopback" and info["ranks_seen_by_rccl"] == world
        # one all-gather per image side and one per match (+ the agreed buffer growth of the first calls)
        assert info["all_gather_calls_rank0"] == 1 + 1 + 2 * 3 + info["agreement_collectives"], info
    ia.free(); ib.free()


@pytest.mark.parametrize("world", [2, 3])
def test_rccl_branch_with_an_in_process_stand_in(ctx, modsx, small_pair, world):
    """The branch a multi-GPU run takes (`modsx_comm_create` with an RCCL id, `transport_all_gather` through librccl's function table,
    its issue lock and abort rules) with W > 1 -- which a one-GPU box cannot do with the real library: `modsx_debug_mock_rccl` fills the
    table with an in-process stand-in.  The single-pair call and a batched call of three pairs (one exchange for all image sides, row-split
    matching, one result all-gather) against the unsharded result, on every rank."""
    from mods_amd import distributed as D
    a, b, _ = small_pair
    views = _views(modsx)
    par = modsx.default_pair_params(ransac_seed=4)
    ia, ib = ctx.upload(a), ctx.upload(b)
    ref = ctx.match_pair_views(ia, ib, views, par)

    def rank_body(r, comm):
        one = comm.ctxs[0].match_pair_views_sharded(comm.comm, ia, ib, views, par, -1)
        many = comm.match_pairs_views_sharded(0, [ia, ia, ia], [ib, ib, ib], views, par, owner_base=-1)
        return one, many, comm.describe()

    out = D.run_mock_rccl(world, rank_body)
    for r, (one, many, info) in enumerate(out):
        _same_pair_result(one, ref)
        assert len(many) == 3
        for m in many:
            _same_pair_result(m, ref)
        assert info["transport"] == "rccl" and info["ranks_seen_by_rccl"] == world and info["all_gather_calls_rank0"] > 0, info
    ia.free(); ib.free()


@pytest.mark.parametrize("world,ndesc,fmt", [(2, 1, 0), (3, 2, 0), (8, 1, 0), (2, 1, 1), (3, 2, 1), (8, 2, 1)])
def test_device_pack_and_order_kernels_equal_the_host_wire_format(ctx, modsx, small_pair, world, ndesc, fmt):
    """k_pack_rows / k_unpack_blocks against modsx_shard_block_pack / modsx_shard_blocks_unpack (the host statement the gloo CPU
    tests run ranks over), for both row formats (fmt 0: whole regions, 200 B; fmt 1: the 56-byte verification slice the batched
    pair call moves): the rows a rank packs are the same bytes, and the device-side ordering of `world` gathered blocks gives the
    same regions, descriptors (every class) and matcher positions as the host's."""
    a = small_pair[0]
    views = _views(modsx)
    par = modsx.default_pair_params(ransac_seed=4)
    ia = ctx.upload(a)
    nv = len(views)
    per_view = [ctx.detect_describe_views(ia, views, par, view_begin=v, view_step=nv) for v in range(nv)]
    ia.free()
    items = 2 * nv                        # two images' worth of items (the second: the views in reverse order)
    blocks = per_view + per_view[::-1]
    RB = modsx.shard_region_bytes(fmt)
    assert RB == (200, 56)[fmt]
    rows = 0
    packs = []
    for r in range(world):
        mine = list(range(r, items, world))
        regs = np.concatenate([blocks[f][0] for f in mine])
        d0 = np.concatenate([blocks[f][1] for f in mine]).astype(np.uint8)
        descs = [d0] + [np.ascontiguousarray(255 - d0)] * (ndesc - 1)
        cnt = np.zeros(items, np.int32)
        for f in mine:
            cnt[f] = len(blocks[f][0])
        packs.append((regs, descs, cnt))
        rows = max(rows, len(regs))
    allb = []
    hdrB = modsx.shard_block_bytes(items, 0, ndesc, fmt)
    assert modsx.shard_block_bytes(items, rows, ndesc, fmt) == hdrB + rows * (RB + 128 * ndesc)
    for regs, descs, cnt in packs:
        blk = modsx.shard_block_pack(regs, descs, cnt, rows, row_format=fmt)
        dev_rows = ctx.shard_device_pack(regs, descs, row_format=fmt)
        host_rows = blk[hdrB:].reshape(rows, -1)[:len(regs)]
        if fmt == 0:
            # field-wise for the region part (struct padding is not data), bytes for the descriptors
            assert same_records(np.frombuffer(np.ascontiguousarray(dev_rows[:, :200]).tobytes(), modsx.REGION),
                                np.frombuffer(np.ascontiguousarray(host_rows[:, :200]).tobytes(), modsx.REGION))
        else:
            # the slice holds no padding: bytes, and they are the seven geometry doubles of reproj_kp
            assert np.array_equal(dev_rows[:, :RB], host_rows[:, :RB])
            assert np.array_equal(np.frombuffer(np.ascontiguousarray(host_rows[:, :RB]).tobytes(), np.float64).reshape(-1, 7),
                                  modsx.shard_kp_rows(regs))
        assert np.array_equal(dev_rows[:, RB:], host_rows[:, RB:])
        allb.append(blk)
    allb = np.concatenate(allb)
    h_regs, h_descs, h_cnt = modsx.shard_blocks_unpack(allb, world, items, rows, ndesc, row_format=fmt)
    d_regs, d_descs, d_pos = ctx.shard_device_unpack(allb, world, items, rows, ndesc, row_format=fmt)
    n = len(h_regs)
    assert n == sum(len(b[0]) for b in blocks) and np.array_equal(h_cnt, [len(b[0]) for b in blocks])
    for k in range(ndesc):
        assert np.array_equal(d_descs[k][:n], h_descs[k])
    if fmt == 1:
        assert np.array_equal(d_regs[:n], h_regs) and np.array_equal(h_regs, modsx.shard_kp_rows(np.concatenate([b[0] for b in blocks])))
        assert np.array_equal(d_pos[:n], h_regs[:, :2])
        return
    assert same_records(d_regs[:n], h_regs)
    assert np.array_equal(d_pos[:n, 0], h_regs["reproj_kp"]["x"]) and np.array_equal(d_pos[:n, 1], h_regs["reproj_kp"]["y"])


@pytest.mark.parametrize("world,npairs", [(2, 3), (3, 5), (8, 4), (8, 1)])
def test_batched_pairs_one_exchange_equals_single_pairs(ctx, modsx, small_pair, world, npairs):
    """modsx_match_pairs_views_sharded: the views of all 2 n images travel in ONE exchange (items = (image, view), item f on rank
    f mod world).  owner_base = 1: pair g is matched and verified on rank (1 + g) mod world alone -- ONE collective per call, the
    other ranks carry the region counts; owner_base = -1: every rank returns every pair (row-split matching, one result
    all-gather more).  Every pair's result == modsx_match_pair_views of that pair (distinct pairs: the two images swapped, a
    cropped pair, ...) whatever n."""
    from mods_amd import distributed as D
    a, b, _ = small_pair
    imgs = [a, b, a[:200, :300].copy(), b[:200, :300].copy(), np.ascontiguousarray(a[::-1]), b]
    dev = [ctx.upload(x) for x in imgs]
    pairs = [(0, 1), (1, 0), (2, 3), (4, 5), (0, 5)][:npairs]
    views = _views(modsx)
    par = modsx.default_pair_params(ransac_seed=4)
    refs = [ctx.match_pair_views(dev[i], dev[j], views, par) for i, j in pairs]

    def rank_body(r, comm):
        before = comm.describe()["all_gather_calls_rank0"]
        out = comm.match_pairs_views_sharded(0, [dev[i] for i, _ in pairs], [dev[j] for _, j in pairs], views, par, owner_base=1)
        out2 = comm.match_pairs_views_sharded(0, [dev[i] for i, _ in pairs], [dev[j] for _, j in pairs], views, par, owner_base=-1)
        info = comm.describe()
        return out, out2, info["all_gather_calls_rank0"] - before - info["agreement_collectives"]

    for r, (out, out2, ncoll) in enumerate(D.run_loopback(world, rank_body)):
        assert ncoll <= 1 + 2 + 1, ncoll                     # owner call: one exchange; all-ranks call: exchange + result all-gather (+ a block retry at most)
        for g, ref in enumerate(refs):
            _same_pair_result(out2[g], ref)
            if (1 + g) % world == r:
                _same_pair_result(out[g], ref)
            else:
                assert out[g]["n_regions"] == ref["n_regions"] and out[g]["n_tentatives"] == 0 and out[g]["n_verified"] == 0
    for d in dev:
        d.free()


@pytest.mark.parametrize("world,npairs,transport", [(2, 3, "loopback"), (3, 5, "loopback"), (8, 4, "loopback"), (8, 1, "loopback"),
                                                     (2, 3, "mock_rccl"), (3, 4, "mock_rccl")])
def test_owner_only_exchange_equals_the_all_gather(ctx, modsx, small_pair, world, npairs, transport):
    """MODSX_EXCHANGE_OWNER: the rows of pair g travel to rank (owner_base + g) mod world alone (headers to every rank, then one group
    of sends and receives -- ncclSend / ncclRecv on the RCCL branch, here through the in-process stand-in) instead of to every rank.
    Every owner's result == modsx_match_pair_views of that pair, the other ranks carry the region counts, two calls in a row (the
    second without any growth), and a rank receives only its pairs' rows: the bytes of the point-to-point transfers summed over the
    ranks are the rows of all pairs ONCE (the all-gather moves them `world` times, padded)."""
    from mods_amd import distributed as D
    a, b, _ = small_pair
    imgs = [a, b, a[:200, :300].copy(), b[:200, :300].copy(), np.ascontiguousarray(a[::-1]), b]
    dev = [ctx.upload(x) for x in imgs]
    pairs = [(0, 1), (1, 0), (2, 3), (4, 5), (0, 5)][:npairs]
    views = _views(modsx)
    par = modsx.default_pair_params(ransac_seed=4)
    refs = [ctx.match_pair_views(dev[i], dev[j], views, par) for i, j in pairs]

    def rank_body(r, comm):
        i1, i2 = [dev[i] for i, _ in pairs], [dev[j] for _, j in pairs]
        ag = comm.match_pairs_views_sharded(0, i1, i2, views, par, owner_base=1)       # the default exchange first
        st0 = modsx.comm_stats(comm.comm)
        comm.set_exchange(modsx.EXCHANGE_OWNER)
        out = comm.match_pairs_views_sharded(0, i1, i2, views, par, owner_base=1)
        st1 = modsx.comm_stats(comm.comm)
        out_b = comm.match_pairs_views_sharded(0, i1, i2, views, par, owner_base=0)
        st2 = modsx.comm_stats(comm.comm)
        every = comm.match_pairs_views_sharded(0, i1, i2, views, par, owner_base=-1)   # every rank wants every pair: stays an all-gather
        st3 = modsx.comm_stats(comm.comm)
        return ag, out, out_b, every, st0, st1, st2, st3

    res = (D.run_loopback if transport == "loopback" else D.run_mock_rccl)(world, rank_body)
    row_b = 56 + 128                                                            # MODSX_SHARD_ROW_KP, one descriptor class
    total_rows = sum(ref["n_regions"][0] + ref["n_regions"][1] for ref in refs)
    for r, (ag, out, out_b, every, st0, st1, st2, st3) in enumerate(res):
        for g, ref in enumerate(refs):
            _same_pair_result(every[g], ref)
            for got, base in ((ag, 1), (out, 1), (out_b, 0)):
                if (base + g) % world == r:
                    _same_pair_result(got[g], ref)
                else:
                    assert got[g]["n_regions"] == ref["n_regions"] and got[g]["n_tentatives"] == 0 and got[g]["n_verified"] == 0
        assert st0["exchanges"] == 0 and st1["exchanges"] == 1 and st2["exchanges"] == 2 and st3["exchanges"] == 2
        assert st2["agreements"] == st1["agreements"]                           # the second owner-only call grows nothing
        mine = sum(refs[g]["n_regions"][0] + refs[g]["n_regions"][1] for g in range(len(refs)) if (1 + g) % world == r)
        assert st1["bytes_received"] - st0["bytes_received"] == mine * row_b, (r, st0, st1, mine)
        # the headers still go to every rank: one small all-gather per owner-only call (plus agreed growth)
        assert st1["bytes_gathered"] - st0["bytes_gathered"] < 64 * 1024 * world
    assert sum(x[5]["bytes_received"] - x[4]["bytes_received"] for x in res) == total_rows * row_b
    for d in dev:
        d.free()


@pytest.mark.parametrize("world", [2, 3])
def test_owner_only_exchange_with_alternating_descriptor_lists(ctx, modsx, small_pair, world):
    """One communicator, one lane, calls that alternate between one descriptor class and two (RootSIFT, then RootSIFT + HalfRootSIFT as
    the WxBS steps carry them, then one again): the owner-only buffers are agreed in ROWS, and a row of a two-class call is 128 bytes
    wider -- with row counts under the caps of the first call the second must still fit (round-5 advisor finding: the buffers were sized
    for the rows of the call that grew them).  Every owner's result == the unsharded call with the same parameters."""
    from mods_amd import distributed as D
    a, b, _ = small_pair
    dev = [ctx.upload(x) for x in (a, b)]
    views = _views(modsx)
    pars = [modsx.default_pair_params(ransac_seed=4),
            modsx.default_pair_params(ransac_seed=4, descs=[(1, 0.8), (3, 0.8)]),
            modsx.default_pair_params(ransac_seed=4)]
    pairs = [(0, 1), (1, 0), (0, 1)]
    refs = [[ctx.match_pair_views(dev[i], dev[j], views, par) for i, j in pairs] for par in pars]

    def rank_body(r, comm):
        comm.set_exchange(modsx.EXCHANGE_OWNER)
        i1, i2 = [dev[i] for i, _ in pairs], [dev[j] for _, j in pairs]
        return [comm.match_pairs_views_sharded(0, i1, i2, views, par, owner_base=0) for par in pars]

    res = D.run_loopback(world, rank_body)
    for r, outs in enumerate(res):
        for k, out in enumerate(outs):
            for g, ref in enumerate(refs[k]):
                if g % world == r:
                    _same_pair_result(out[g], ref)
                else:
                    assert out[g]["n_regions"] == ref["n_regions"] and out[g]["n_tentatives"] == 0
    for d in dev:
        d.free()


def test_owner_only_exchange_errors_are_collective(ctx, modsx, small_pair):
    """In the owner-only mode a rank-local failure still travels in the header every rank receives (same error, same call, the
    communicator keeps working), and ranks set to DIFFERENT modes do not hang: the mode is part of the header's magic."""
    from mods_amd import distributed as D
    a, b, _ = small_pair
    good = _views(modsx)[:4]
    bad = list(good)
    bad[1] = modsx.make_view(1e9, 0.0, 1.0, 0.2, 1)
    par = modsx.default_pair_params(ransac_seed=4)
    ia, ib = ctx.upload(a), ctx.upload(b)
    ref = ctx.match_pair_views(ia, ib, good, par)

    def rank_body(r, comm):
        comm.set_exchange(modsx.EXCHANGE_OWNER)
        try:
            comm.match_pairs_views_sharded(0, [ia, ia], [ib, ib], bad, par, owner_base=0)
            err = None
        except RuntimeError as e:
            err = str(e)
        out = comm.match_pairs_views_sharded(0, [ia, ia], [ib, ib], good, par, owner_base=0)
        return err, out

    for r, (err, out) in enumerate(D.run_loopback(2, rank_body)):
        assert err is not None and ("degenerate" in err if r == 1 else "rank 1" in err), (r, err)
        _same_pair_result(out[r], ref)

    def mixed(r, comm):
        if r == 0:
            comm.set_exchange(modsx.EXCHANGE_OWNER)
        t0 = time.time()
        try:
            comm.match_pairs_views_sharded(0, [ia], [ib], good, par, owner_base=0)
            return None, time.time() - t0
        except RuntimeError as e:
            return str(e), time.time() - t0

    for err, dt in D.run_loopback(2, mixed, timeout_ms=3000):
        assert err is not None and dt < 20, (err, dt)
    ia.free(); ib.free()


def test_sharded_block_retry_and_unbalanced_ranks(ctx, modsx, small_pair, monkeypatch):
    """A first block size far below a rank's row count: every rank sees the overflow in the gathered headers, all grow alike
    and repeat the exchange (and agree on the allocation)."""
    from mods_amd import distributed as D
    monkeypatch.setenv("MODSX_SHARD_BLOCK_ROWS", "7")
    a, b, _ = small_pair
    views = _views(modsx)
    par = modsx.default_pair_params(ransac_seed=4)
    ia = ctx.upload(a)
    ref1, refd1, c1 = ctx.detect_describe_views(ia, views, par, want_counts=True)

    def rank_body(r, comm):
        r1, d1ptr, cnt = comm.detect_describe_views_sharded(0, ia, views, par)
        r1b, d1ptrb, _ = comm.detect_describe_views_sharded(0, ia, views, par)      # second call: block size is known
        return r1, dev_to_host(d1ptr, len(r1) * 128).reshape(-1, 128), r1b, comm.describe()

    for r1, desc, r1b, info in D.run_loopback(3, rank_body):
        assert same_records(r1, ref1) and same_records(r1b, ref1)
        assert np.array_equal(desc, refd1.astype(np.uint8))
        assert info["block_retries"] == 1 and info["agreement_collectives"] == 2
    ia.free()


def test_sharded_lanes_two_contexts_per_rank(ctx, modsx, small_pair):
    """Two contexts per rank drive different pairs at the same time over ONE communicator: the round-robin lane order keeps
    the collectives of the ranks in one sequence."""
    import threading
    from mods_amd import distributed as D
    a, b, _ = small_pair
    views = _views(modsx)[:5]
    par = modsx.default_pair_params(ransac_seed=2)
    ia, ib = ctx.upload(a), ctx.upload(b)
    refs = [ctx.match_pair_views(ia, ib, views, par), ctx.match_pair_views(ib, ia, views, par)]

    def rank_body(r, comm):
        res = [None, None]

        def lane(w):
            x, y = (ia, ib) if w == 0 else (ib, ia)
            outs = []
            for k in range(3):
                outs.append(comm.ctxs[w].match_pair_views_sharded(comm.comm, x, y, views, par, -1))
                if w == 1 and k == 0:
                    time.sleep(0.05)     # lanes out of phase
            res[w] = outs
        th = [threading.Thread(target=lane, args=(w,)) for w in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return res

    for res in D.run_loopback(2, rank_body, lanes=2):
        for w in range(2):
            for got in res[w]:
                _same_pair_result(got, refs[w])
    ia.free(); ib.free()


def test_sharded_error_is_collective(ctx, modsx, small_pair):
    """A view only rank 1 owns is degenerate: rank 1's failure travels in its block header, every rank returns the same
    error from the same call, and the communicator keeps working."""
    from mods_amd import distributed as D
    a, b, _ = small_pair
    good = _views(modsx)[:4]
    bad = list(good)
    bad[1] = modsx.make_view(1e9, 0.0, 1.0, 0.2, 1)
    par = modsx.default_pair_params(ransac_seed=4)
    ia = ctx.upload(a)
    ref1, _ = ctx.detect_describe_views(ia, good, par)

    def rank_body(r, comm):
        t0 = time.time()
        try:
            comm.detect_describe_views_sharded(0, ia, bad, par)
            err = None
        except RuntimeError as e:
            err = str(e)
        r1, _, _ = comm.detect_describe_views_sharded(0, ia, good, par)
        return err, time.time() - t0, r1

    for r, (err, dt, r1) in enumerate(D.run_loopback(2, rank_body)):
        assert err is not None and ("degenerate" in err if r == 1 else "rank 1" in err), (r, err)
        assert dt < 20
        assert same_records(r1, ref1)
    ia.free()


def test_sharded_watchdog_turns_a_missing_rank_into_an_error(ctx, modsx, small_pair):
    """Rank 1 never makes the call: rank 0's collective hits the deadline, the communicator is aborted, and the call -- and
    every later one -- returns MODSX_ERR_TIMEOUT instead of hanging."""
    from mods_amd import distributed as D
    a, _, _ = small_pair
    views = _views(modsx)[:3]
    par = modsx.default_pair_params()
    ia = ctx.upload(a)

    def rank_body(r, comm):
        if r == 1:
            return None
        out = []
        for _ in range(2):
            t0 = time.time()
            try:
                comm.detect_describe_views_sharded(0, ia, views, par)
                out.append((None, time.time() - t0))
            except RuntimeError as e:
                out.append((str(e), time.time() - t0))
        return out, comm.describe()

    res = D.run_loopback(2, rank_body, timeout_ms=1500)
    (e1, t1), (e2, t2) = res[0][0]
    assert e1 is not None and "dead" in e1 and 1.0 < t1 < 15.0, (e1, t1)
    assert e2 is not None and "dead" in e2 and t2 < 1.0, (e2, t2)
    ia.free()


def test_sharded_ladder_equals_unsharded_ladder(ctx, modsx, small_pair):
    """configs[3] in miniature over 3 ranks: an MSER step and two HessianAffine steps with cumulative view sets; all steps
    forced, and the early exit taken at the same step on every rank."""
    from mods_amd import distributed as D
    a, b, _ = small_pair
    prev = {0: [], 3: []}
    steps = []
    for det, scales, tilts, phi, sigma, ratio in ((3, [1, 0.5], [1], 360.0, 0.8, 0.85), (0, [1], [1, 2, 4], 360.0, 0.2, 0.8),
                                                 (0, [1], [1, 2, 4], 120.0, 0.2, 0.8)):
        v = modsx.set_vs_pars(scales, tilts, phi, sigma, 1, prev[det])
        assert v
        steps.append((v, ratio, det))
    par = modsx.default_pair_params(ransac_seed=3)
    ia, ib = ctx.upload(a), ctx.upload(b)
    ref_all, done_all = ctx.match_ladder(ia, ib, steps, par, min_matches=10 ** 6)
    ref_early, done_early = ctx.match_ladder(ia, ib, steps, par, min_matches=10)
    assert done_all == 3

    def rank_body(r, comm):
        return (comm.match_ladder_sharded(0, ia, ib, steps, par, min_matches=10 ** 6),
                comm.match_ladder_sharded(0, ia, ib, steps, par, min_matches=10))

    for (got_all, d_all), (got_early, d_early) in D.run_loopback(3, rank_body):
        assert d_all == done_all and d_early == done_early
        _same_pair_result(got_all, ref_all)
        _same_pair_result(got_early, ref_early)
    ia.free(); ib.free()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_wxbs_ladder_two_descriptor_classes(ctx, modsx, small_pair, world):
    """The WxBS step structure through the exchange: a block row carries the region and BOTH descriptors of the step (200 +
    2 x 128 B), the device-side ordering fills one accumulator per (detector, descriptor) class, and every class is matched
    with its query rows split over the ranks; the result on every rank == the unsharded ladder (which test_gpu_views.py
    compares with the oracle)."""
    from mods_amd import distributed as D
    from test_gpu_views import _wxbs_ladder, _wxbs_ladder_params
    a, b, _ = small_pair
    _, steps = _wxbs_ladder(None, modsx, which=(0, 2, 3))
    par = _wxbs_ladder_params(modsx, 7, 0, 300, 120)
    ia, ib = ctx.upload(a), ctx.upload(b)
    ref, done = ctx.match_ladder(ia, ib, steps, par, min_matches=10 ** 6)
    assert done == 3 and ref["n_verified"] > 20

    def rank_body(r, comm):
        return comm.match_ladder_sharded(0, ia, ib, steps, par, min_matches=10 ** 6)

    for got, d in D.run_loopback(world, rank_body):
        assert d == 3
        _same_pair_result(got, ref)
    ia.free(); ib.free()


def test_configs3_full_cviu_ladder_sharded_over_8_ranks(ctx, modsx):
    """configs[3] with its view sharding: every step of the iters_mods_cviu.ini ladder (27 MSER + 61 HessianAffine views per
    image, all steps forced) on the 1024x768 pair with the views of each step split over 8 ranks; every rank ends with the
    result of the unsharded ladder (which test_gpu_views.py compares with the oracle)."""
    from mods_amd import distributed as D, synthetic
    from test_gpu_views import _cviu_ladder
    a, b, _ = synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345)
    _, steps = _cviu_ladder(None, modsx)
    par = modsx.default_pair_params(ransac_seed=3, ori_mrSize=5.1962)
    ia, ib = ctx.upload(a), ctx.upload(b)
    ref, done = ctx.match_ladder(ia, ib, steps, par, min_matches=10 ** 6)
    assert done == 5 and ref["n_regions"][0] > 50000

    def rank_body(r, comm):
        return comm.match_ladder_sharded(0, ia, ib, steps, par, min_matches=10 ** 6), comm.describe()

    for got, info in D.run_loopback(8, rank_body):
        assert got[1] == 5
        _same_pair_result(got[0], ref)
        assert info["ranks_seen_by_rccl"] == 8 and info["all_gather_calls_rank0"] >= 5 * 3
    ia.free(); ib.free()

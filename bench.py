#!/usr/bin/env python3
"""Benchmark of the MODS hot path on MI355X (BASELINE.json: image-pairs/s + descriptors matched/s, 1024x768, HessAff+SIFT).

A step = one pass of the path over one batch of 64 synthetic 1024x768 image pairs per GPU (16 contexts take the pairs of a
step from a shared counter): per image the affine view ladder
(default TiltSet 1,2,4,6,8 at Phi 120 = 31 views, synth-detection.cpp:103-234) -> Hessian-Affine + Baumberg -> dominant
orientation -> RootSIFT per view, then brute-force FGINN matching of the ~24 k x 24 k descriptors on the int8 matrix cores
(matching.cpp:357-461), duplicate filtering and LO-RANSAC (H).  All images are resident in HBM before the timed region.

  --config views31 (default) | views61 | views11 | views8 | views1 | wxbs      workload of the JSON line
  --gpus N   one process per GPU (torch.distributed.run).  --shard pairs (default): independent pairs per rank, no
             collective.  --shard views: every pair's views are split over the N ranks (view v -> rank v mod N), the region
             rows + u8 descriptors are all-gathered over RCCL/xGMI inside the library (modsx_detect_describe_views_sharded)
             and the query rows of the match are split over the ranks (modsx_match_fginn_sharded); the batch grows with N
             (weak scaling: per-GPU work fixed).
Rank 0 prints ONE JSON line: value = pairs/s of the whole job, `roofline` = the distance kernels (all k_match_*
launches, sweep 2 included, on the descriptors of this run) against the int8 MFMA peak, `roofline_describe` = the
describe stage against HBM with SURVEY section 8(d) bytes, `cpu_baseline` = the CPU oracle (restatement of the reference's
CPU path) on the same workload with all host cores and with one, `parity` = GPU vs that CPU path on one pair of the run.
"""
import argparse
import itertools
import threading
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
INT8_PEAK_TOPS = 5000.0      # dense int8 matrix peak (~2x the 2.5 PF bf16 dense peak)
CONFIGS = {   # name -> (tilts, phi, detector mode, description)
    "views1": ("1", 360.0, "configs[1]: 1 view"),
    "views8": ("1,2,3,4,6", 360.0, "configs[2]: 8 affine-synth views (TiltSet 1,2,3,4,6, Phi 360)"),
    "views11": ("1,2,4,6,8", 360.0, "11 views (TiltSet 1,2,4,6,8, Phi 360: iters_mods_cviu.ini HessianAffine step)"),
    "views31": ("1,2,4,6,8", 120.0, "31 views (TiltSet 1,2,4,6,8, Phi 120)"),
    "views61": ("1,2,4,6,8", 60.0, "61 views (TiltSet 1,2,4,6,8, Phi 60)"),
    # configs[4]: 1920x1080 pairs with the parameter set of config_iter_mods_cviu_wxbs.ini (NotLessThanRegions 2000, maxAngles 5
    # on the 5.1962 region, HalfRootSIFT, contradDist 10, duplicateDist 3, err_threshold 4, max_samples 1e6), H then F
    "wxbs": ("1", 360.0, "configs[4]: WxBS parameter set, identity view"),
}
WXBS = dict(mode=4, threshold=5.3333, reg_number=2000, ori_mrSize=5.1962, ori_maxAngles=5, ori_threshold=0.8, desc_mrSize=5.1962,
            desc_photoNorm=1, desc_type=3, desc_maxBinValue=0.2, match_ratio=0.8, contradDist=10.0, duplicateDist=3.0,
            err_threshold=4.0, confidence=0.99, max_samples=1000000, localOptimization=1, LAFCoef=3.0, HLAFCoef=13.0, doSymmCheck=1)


def cpu_info():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def oracle_pair_views(O, a, b, views_o, params, threads, seed=1, query_cap=None, one_image=False):
    """The CPU path (oracle = restatement of the reference) for one pair under a view ladder.  Parallel structure of the
    reference: images x views in parallel (mods.cpp:255-271, imagerepresentation.cpp:612-622), here a thread pool over
    (image, view) tasks (ctypes releases the GIL) + OpenMP over the query rows of the brute-force kNN."""
    import numpy as np
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import laf_of
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
    except OSError:
        pass
    t0 = time.time()
    imgs = [a] if one_image else [a, b]

    def one(task):
        im, v = task
        return O.detect_describe_views(imgs[im], [views_o[v]])
    tasks = [(i, v) for i in range(len(imgs)) for v in range(len(views_o))]
    if threads > 1:
        with ThreadPoolExecutor(threads) as pool:
            parts = list(pool.map(one, tasks))
    else:
        parts = [one(t) for t in tasks]
    regs, desc = [], []
    for i in range(len(imgs)):
        rs, ds, size = [], [], 0
        for v in range(len(views_o)):
            r, d = parts[i * len(views_o) + v]
            r = r.copy()
            if v and len(r):      # the per-view call numbered its block from 0 and cannot know the view index
                ident = abs(views_o[v].tilt - 1) <= 0.1 and abs(views_o[v].phi) <= 0.2 and abs(views_o[v].zoom - 1) <= 0.1
                r["img_id"] = 0 if ident else v
            r["id"] += size; r["parent_id"] += size
            size += len(r)
            rs.append(r); ds.append(d)
        regs.append(np.concatenate(rs)); desc.append(np.concatenate(ds))
    t1 = time.time()
    out = dict(t_extract=t1 - t0, regs=regs, desc=desc)
    if one_image:
        return out
    r1, r2, d1, d2 = regs[0], regs[1], desc[0], desc[1]
    pos2 = np.stack([r2["reproj_kp"]["x"], r2["reproj_kp"]["y"]], 1)
    nq = len(d1) if query_cap is None else min(query_cap, len(d1))
    tent = O.match_fginn(d1[:nq], d2, pos2, params.match_ratio, params.contradDist)
    t2 = time.time()
    out.update(t_match=t2 - t1, n_match_queries=nq, tent=tent)
    if query_cap is not None and nq < len(d1):
        return out
    pts = np.stack([r1["reproj_kp"]["x"][tent["q"]], r1["reproj_kp"]["y"][tent["q"]],
                    r2["reproj_kp"]["x"][tent["t0"]], r2["reproj_kp"]["y"][tent["t0"]]], 1)
    order, keep = O.duplicate_filtering(pts, tent["ratio"], params.duplicateDist, True)
    sel = order[keep]
    tu, pu = tent[sel], pts[sel]
    res = None
    if O.ref_available():
        res = O.loransac_h(pu, laf_of(r1, tu["q"]), laf_of(r2, tu["t0"]), err_threshold=params.err_threshold,
                           confidence=params.confidence, max_samples=params.max_samples, seed=seed)
    t3 = time.time()
    out.update(t_verify=t3 - t2, uniq=tu, pts=pu, ransac=res, total=t3 - t0, laf1=laf_of(r1, tu["q"]), laf2=laf_of(r2, tu["t0"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=768)
    ap.add_argument("--cols", type=int, default=1024)
    ap.add_argument("--config", type=str, default="views31", choices=sorted(CONFIGS))
    ap.add_argument("--tilts", type=str, default="", help="override the tilt set of --config, e.g. 1,2,3,4,6")
    ap.add_argument("--phi", type=float, default=0.0)
    ap.add_argument("--batch", type=int, default=0, help="pairs per step per GPU (default 64; 256 for views1, 128 for wxbs)")
    ap.add_argument("--blobs", type=int, default=0, help="blobs per 1024x768 of the synthetic scene (0 = per config)")
    ap.add_argument("--workers", type=int, default=16, help="contexts (thread + stream) per GPU")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic pairs cycled through the batch")
    ap.add_argument("--init-sigma", type=float, default=0.2)
    ap.add_argument("--shard", type=str, default="", choices=["", "pairs", "views"])
    ap.add_argument("--batch-api", action="store_true", help="multi-view configs: run a step as ONE modsx_match_pairs_views call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short 1/8/11-view runs reported under `extra`")
    args = ap.parse_args()

    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    # default: pairs over ranks (no collective: independent pairs are the reference's batch use).  --shard views is the
    # north-star's view-parallel mode; it needs RCCL between the ranks and is opt-in because the 1-GPU development box
    # cannot exercise world > 1 (the path is covered at world 1 on the GPU and at world 2/3 over gloo on recorded blocks).
    shard = args.shard or "pairs"

    import mods_amd
    from mods_amd import synthetic
    tilts, phi, cfg_desc = CONFIGS[args.config]
    wxbs = args.config == "wxbs"
    if wxbs:
        args.rows, args.cols = 1080, 1920
    if args.tilts:
        tilts, phi = args.tilts, (args.phi or 360.0)
        cfg_desc = "tilts %s, phi %g" % (tilts, phi)
    single_view = tilts == "1"
    # a step ends with the verification of its last pairs while the GPU drains: configs[1] (45 ms per 64 pairs) and
    # configs[4] take longer steps so that this tail stays a few per cent of the step
    batch = args.batch or (128 if wxbs else 256 if single_view else 64)
    # blob density of the synthetic scene: 4000 per 1024x768 for configs[1] (comparable with round 1), 5500 for the
    # multi-view configs so that the 31-view default carries the >= 50 k descriptors per pair the north star is quoted on
    blobs = args.blobs or (4000 if (single_view or wxbs) else 5500)
    nblobs = int(blobs * args.rows * args.cols / (768.0 * 1024))
    ctxs = [mods_amd.Context(local_rank) for _ in range(max(1, args.workers))]
    ctx = ctxs[0]
    params = mods_amd.default_pair_params(ransac_seed=1, **(WXBS if wxbs else {}))
    views = mods_amd.set_vs_pars([1.0], [float(t) for t in tilts.split(",")], phi, args.init_sigma, 1, [])

    # pairs: with --shard views every rank holds every pair of the (N x larger) batch; otherwise its own pairs
    comm = None
    if shard == "views":
        from mods_amd import distributed as D
        comm = D.NativeComm(ctxs, dist)          # one RCCL communicator per context (stream), bootstrapped over torch
        seed0, nbatch = 12345, batch * world
    else:
        seed0, nbatch = 12345 + 1000 * rank, batch
    pairs_host = [synthetic.make_pair(rows=args.rows, cols=args.cols, nblobs=nblobs, seed=seed0 + 17 * i)
                  for i in range(max(1, args.distinct))]
    Hgt = pairs_host[0][2]
    dev = [(ctx.upload(a), ctx.upload(b)) for a, b, _ in pairs_host]
    imgs1 = [dev[i % len(dev)][0] for i in range(nbatch)]
    imgs2 = [dev[i % len(dev)][1] for i in range(nbatch)]

    def barrier():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(len(ctxs))

    def run_batch(vw=views, single=single_view, i1=imgs1, i2=imgs2, cx=ctxs):
        if single:
            return mods_amd.match_pairs(cx, i1, i2, params)
        if comm is None and args.batch_api:
            # the library's own batch loop (modsx_match_pairs_views): 4-5 % below the persistent python threads of the default
            # path, whose threads are not re-created per step
            return mods_amd.match_pairs_views(cx, i1, i2, vw, params, arrays=False)

        # multi-view pairs: one python thread per context (ctypes releases the GIL inside the library).  The contexts take the
        # pairs of the step from a shared counter, so that the step ends at most one pair after its last pair was started
        # (with a static pair -> context map the step waited for the context with the most expensive pairs: 15 % of the
        # GPU's time was idle at step boundaries).  The view-sharded mode keeps the static map: every rank must meet the
        # others' collectives in the same order.
        nxt = itertools.count()
        lock = threading.Lock()

        def work(w):
            out = []
            if comm is not None:
                for i in range(w, len(i1), len(cx)):
                    out.append((i, comm.match_pair_views_sharded(w, i1[i], i2[i], vw, params, owner=i % world)))
                return out
            while True:
                with lock:
                    i = next(nxt)
                if i >= len(i1):
                    return out
                out.append((i, cx[w].match_pair_views(i1[i], i2[i], vw, params)))
        res = [None] * len(i1)
        for part in pool.map(work, range(len(cx))):
            for i, r in part:
                res[i] = r
        return res

    for _ in range(args.warmup):
        results = run_batch()
    barrier()
    t0 = time.perf_counter()
    ndesc = 0
    for _ in range(args.steps):
        results = run_batch()
        for r in results:
            if r is not None:
                ndesc += r["n_regions"][0] + r["n_regions"][1]
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    res = next(r for r in results if r is not None)
    verify_timed = mods_amd.last_batch_verify() if single_view else None   # host share of the last batch of the timed region

    # ---- untimed legs (rank 0 reports them) -------------------------------------------------------------------------
    # per-kernel time of the multi-stream regime: two extra steps with the event brackets on
    PROF_STEPS = 1
    stats = {}
    if rank == 0 or comm is not None:
        for c in ctxs:
            c.profile(True)
        for _ in range(PROF_STEPS):
            run_batch()
        if comm is not None:
            barrier()
        else:
            for c in ctxs:
                c.synchronize()
        for c in ctxs:
            for k, v in c.kernel_stats().items():
                d = stats.setdefault(k, dict(ms=0.0, work=0.0, launches=0))
                d["ms"] += v["ms"]; d["work"] += v["work"]; d["launches"] += v["launches"]
            c.profile(False)
    # Roofline leg: in the timed region the kernels of --workers streams time-slice the CUs, so an event pair there
    # brackets queueing as well.  The same pairs are repeated on ONE stream and the launch durations come from that pass.
    iso, niso = {}, 0
    if rank == 0 and (comm is None or world == 1):
        ctx.profile(True)
        niso = min(nbatch, 8 if single_view else 4)
        if single_view:
            for i0 in range(0, niso, 4):
                mods_amd.match_pairs([ctx], imgs1[i0:min(i0 + 4, niso)], imgs2[i0:min(i0 + 4, niso)], params)
        else:
            for i in range(niso):
                ctx.match_pair_views(imgs1[i], imgs2[i], views, params)
        ctx.synchronize()
        iso = {k: v for k, v in ctx.kernel_stats().items() if v["launches"]}
        ctx.profile(False)
    if dist is not None:
        t = torch.tensor([elapsed, float(ndesc)], device="cuda", dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        ndesc_total = float(tsum[1])
    else:
        ndesc_total = float(ndesc)

    if rank == 0:
        pairs = args.steps * (nbatch if comm is not None else world * nbatch)
        value = pairs / elapsed
        out = {
            "metric": "image-pairs/sec (1024x768, HessAff+RootSIFT over the affine view ladder, MFMA FGINN match, LO-RANSAC H)",
            "value": value, "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (detect/describe) + u8 on the int8 matrix cores (distance matrix)", "data": "synthetic",
            "config": {"workload": "%dx%d synthetic pair, %s = %d views per image, HessAff+RootSIFT, MFMA distance matrix + "
                                   "FGINN, duplicate filter, LO-RANSAC H" % (args.cols, args.rows, cfg_desc, len(views)),
                       "views": len(views), "pairs_per_step": nbatch if comm is not None else world * nbatch,
                       "workers_per_gpu": len(ctxs), "distinct_pairs": len(dev),
                       "parallelism": ("views of every pair sharded over %d ranks (v mod N), RCCL all-gather of region rows + u8 "
                                       "descriptors, query rows of the match split over ranks" % world) if comm is not None else
                                      "pairs sharded over ranks, no collective"},
            "descriptors_per_s": ndesc_total / elapsed,
            "descriptors_per_pair": ndesc_total / pairs,
            "result": {"regions": list(res["n_regions"]), "tentatives": res["n_tentatives"], "unique": res["n_unique"],
                       "verified": res["n_verified"],
                       "H_vs_generator_max_abs": float(np.abs(res["H"] / res["H"][2, 2] - Hgt).max())},
            "kernel_ms_per_pair_multi_stream": {k: v["ms"] / (PROF_STEPS * nbatch) for k, v in stats.items() if v["launches"]},
        }
        if comm is not None:
            out["rccl"] = comm.describe()
        if iso:
            per = {k: v["ms"] / niso for k, v in iso.items()}
            out["kernels_single_stream_ms_per_pair"] = per
            m = iso.get("match_fginn")
            if m and m["launches"]:
                n1, n2 = res["n_regions"]
                ms = m["ms"] / m["launches"]
                flops = m["work"] / m["launches"]
                tf = flops / (ms * 1e-3) / 1e12
                byts = (n1 + n2) * 128.0 + n1 * 32.0
                out["roofline"] = {
                    "kernel": "k_match_* (pack, sweep1, decide, sweep2, events: every launch of one matching problem)",
                    "bound": "mfma", "achieved": tf, "peak": INT8_PEAK_TOPS, "unit": "TFLOP/s", "frac": tf / INT8_PEAK_TOPS,
                    "traffic": None, "avg_launch_ms": ms, "algorithmic_work_per_launch": flops, "N": n1, "M": n2,
                    "hbm_view": {"compulsory_bytes": byts, "achieved_GBs": byts / (ms * 1e-3) / 1e9,
                                 "frac_of_8TBs": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    "binds": "mfma: 2*N*M*128 int8 ops against (N+M)*128 B is ~1e4 op/B, the kernel is compute-shaped; "
                             "its limiter is VALU issue beside the MFMAs (DESIGN.md section 5)",
                    "note": "HIP events on the launch stream around the five launches, single-stream pass over %d pairs of "
                            "this run right after the timed region; N, M = regions of the last pair" % niso}
                tfile = os.path.join(ROOT, "profiles", "pmc_traffic_r02.json")
                if os.path.exists(tfile):
                    out["roofline"]["traffic"] = json.load(open(tfile)).get("k_match_total_" + args.config)
            dk = [iso.get(k) for k in ("patch_sample", "blur_rows", "blur_cols", "describe")]
            if all(d and d["launches"] for d in dk):
                ms = sum(d["ms"] for d in dk) / dk[0]["launches"]
                byts = (dk[0]["work"] + dk[3]["work"]) / dk[0]["launches"]
                out["roofline_describe"] = {
                    "kernel": "describe stage: k_sample_rows_lds (sampling fused with the row filter; k_patch_sample + k_patch_blur for the few windows beyond LDS) + k_blur_cols_lds + k_describe (one chunk = one launch of each)",
                    "bound": "hbm", "achieved": byts / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": ms,
                    "algorithmic_work_per_launch": byts, "traffic": None,
                    "note": "bytes per SURVEY section 8(d): (P+2)^2 x 4 B read + 128 B written per region; the limiter of these "
                            "kernels is VALU issue / the texture addresser, not HBM (DESIGN.md section 5)"}
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic_r02.json")
            if "roofline_describe" in out and os.path.exists(tfile):
                t = json.load(open(tfile))
                # HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, rocprofv3 --pmc, separate passes) of one launch of each kernel of the stage
                out["roofline_describe"]["traffic"] = sum(t.get(k, 0) for k in ("k_sample_rows_lds", "k_patch_sample", "k_blur_rows_lds",
                                                                                "k_patch_blur", "k_blur_cols_lds", "k_describe")) or None
        if wxbs:
            # H verification was the timed region; the same batch with epipolar verification, and the host share of both
            vms, vn, vth = verify_timed
            hshare = {"verify_ms_per_pair": vms / max(1, vn), "helper_threads": vth}
            pF = mods_amd.default_pair_params(ransac_seed=1, useF=1, **WXBS)
            mods_amd.match_pairs(ctxs, imgs1, imgs2, pF)
            ta = time.perf_counter()
            stF = 3
            for _ in range(stF):
                rF = mods_amd.match_pairs(ctxs, imgs1, imgs2, pF)
            dtF = time.perf_counter() - ta
            fms, fn, fth = mods_amd.last_batch_verify()
            out["wxbs"] = {
                "H": {"pairs_per_s": value, **hshare,
                      "host_ransac_share_of_wall": (vms / max(1, vn)) / max(1, vth) / (1e3 / value) * 1.0},
                "F": {"pairs_per_s": stF * nbatch / dtF, "verify_ms_per_pair": fms / max(1, fn), "helper_threads": fth,
                      "verified_last_pair": rF[0]["n_verified"],
                      "host_ransac_share_of_wall": (fms / max(1, fn)) / max(1, fth) / (dtF / (stF * nbatch) * 1e3)},
                "note": "DuplicateFiltering + LO-RANSAC run on helper threads beside the device pipeline; share = per-pair verify "
                        "time / helper threads / per-pair wall time (the fraction of the wall the helpers are busy)"}
        # ---- short runs of the other view counts (same timed-region rules, fewer steps) ---------------------------------
        if not args.no_extra and comm is None and world == 1 and not args.tilts and not wxbs:
            extra = {}
            for name in ("views1", "views8", "views11"):
                if name == args.config:
                    continue
                tl, ph, _ = CONFIGS[name]
                vw = mods_amd.set_vs_pars([1.0], [float(t) for t in tl.split(",")], ph, args.init_sigma, 1, [])
                nb = 256 if tl == "1" else 64
                i1 = [dev[i % len(dev)][0] for i in range(nb)]
                i2 = [dev[i % len(dev)][1] for i in range(nb)]
                run_batch(vw, tl == "1", i1, i2)
                for c in ctxs:
                    c.synchronize()
                ta = time.perf_counter()
                nd, st = 0, 3
                for _ in range(st):
                    for r in run_batch(vw, tl == "1", i1, i2):
                        nd += r["n_regions"][0] + r["n_regions"][1]
                for c in ctxs:
                    c.synchronize()
                dt = time.perf_counter() - ta
                extra[name] = {"views": len(vw), "pairs_per_s": st * nb / dt, "descriptors_per_pair": nd / (st * nb),
                               "descriptors_per_s": nd / dt}
            out["extra"] = extra
        # ---- the CPU path on the same workload: parity of one pair of the run + the baseline timing ---------------------
        if world == 1 and not args.no_cpu_baseline and not wxbs:
            from oracle import pyoracle as O
            model, ncpu, usable = cpu_info()
            a, b, _ = pairs_host[0]
            vo = O.set_vs_pars([1.0], [float(t) for t in tilts.split(",")], phi, args.init_sigma, 1, [])
            O.detect_describe_views(a[:64, :64].copy(), vo[:1])     # lazy tables (single-threaded once)
            full = oracle_pair_views(O, a, b, vo, params, threads=usable, seed=1)
            got = ctx.match_pair_views(imgs1[0], imgs2[0], views, params)
            par = {"pair": "pair 0 of the run, GPU path vs CPU oracle",
                   "regions_identical": bool(got["n_regions"] == (len(full["regs"][0]), len(full["regs"][1]))),
                   "tentatives_identical": bool(got["n_unique"] == len(full["uniq"]) and all(
                       np.array_equal(got["tentatives"][f], full["uniq"][f]) for f in ("q", "t0", "tj", "t1", "d1", "d2", "ratio")))}
            if full["ransac"] is not None:
                par["inliers_identical"] = bool(np.array_equal(got["ransac_inlier"], full["ransac"]["inl"]))
                Hc = np.asarray(full["ransac"]["H"], float).reshape(3, 3)
                par["H_vs_cpu_max_abs"] = float(np.abs(got["H"] / got["H"][2, 2] - Hc / Hc[2, 2]).max())
            out["parity"] = par
            # one thread: a bounded sample (image A's views + 2048 query rows of the match), scaled to a full pair
            s1 = oracle_pair_views(O, a, b, vo, params, threads=1, one_image=True)
            try:
                import ctypes
                ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)
            except OSError:
                pass
            nq = min(2048, len(full["desc"][0]))
            pos2 = np.stack([full["regs"][1]["reproj_kp"]["x"], full["regs"][1]["reproj_kp"]["y"]], 1)
            tm = time.time()
            O.match_fginn(full["desc"][0][:nq], full["desc"][1], pos2, params.match_ratio, params.contradDist)
            tm = time.time() - tm
            n1 = len(full["desc"][0])
            est = 2 * s1["t_extract"] + tm * n1 / nq + full["t_verify"]
            base = {"value": 1.0 / full["total"], "unit": "image-pairs/s", "cores": usable, "kind": "port",
                    "cpu_model": model, "host_logical_cpus": ncpu,
                    "sample": "1 full pair of this workload (%d views per image, %d + %d regions) on %d threads in %.2f s: "
                              "(image, view) tasks on a thread pool as the reference's OpenMP loops (mods.cpp:255-271, "
                              "imagerepresentation.cpp:612-622), OpenMP over the query rows of the linear kNN; extract %.2f s, "
                              "match %.2f s, duplicate filter + reference degensac %.2f s"
                              % (len(views), len(full["regs"][0]), len(full["regs"][1]), usable, full["total"],
                                 full["t_extract"], full["t_match"], full["t_verify"]),
                    "descriptors_per_s": (len(full["regs"][0]) + len(full["regs"][1])) / full["total"],
                    "single_thread": {"value": 1.0 / est, "unit": "image-pairs/s", "cores": 1,
                                      "sample": "image A's %d views (%.1f s, doubled) + %d of %d query rows of the match (%.1f s, "
                                                "scaled) + verification of the full pair (%.2f s) -> %.1f s per pair"
                                                % (len(views), s1["t_extract"], nq, n1, tm, full["t_verify"], est)}}
            # the one stage where the real reference runs here: degensac from oracle/_ref vs the product's host C++
            if full["ransac"] is not None and len(full["pts"]) >= 8:
                tr = time.time()
                for _ in range(3):
                    O.loransac_h(full["pts"], full["laf1"], full["laf2"], err_threshold=params.err_threshold,
                                 confidence=params.confidence, max_samples=params.max_samples, seed=1)
                tr = (time.time() - tr) / 3
                tp = time.time()
                for _ in range(3):
                    mods_amd.loransac_h(full["pts"], full["laf1"], full["laf2"], err_threshold=params.err_threshold,
                                        confidence=params.confidence, max_samples=params.max_samples, seed=1)
                tp = (time.time() - tp) / 3
                base["ransac_reference_vs_restatement"] = {
                    "reference_degensac_ms": 1e3 * tr, "libmodsx_host_ms": 1e3 * tp, "ratio": tr / tp if tp > 0 else None,
                    "tentatives": int(len(full["pts"])),
                    "note": "LORANSACFiltering (H) on the tentatives of this pair: the reference's own degensac C sources "
                            "(oracle/_ref, built in the build container) vs the restatement shipped in libmodsx"}
            out["cpu_baseline"] = base
        else:
            out["cpu_baseline"] = None
        # RCCL writes a version banner through C stdio, which would otherwise be flushed at exit, after this line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    for a_, b_ in dev:
        a_.free(); b_.free()
    if comm is not None:
        comm.close()
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the MODS hot path on MI355X.

A step = one pass of the path over one batch of --batch synthetic 1024x768 image pairs, 1 (identity)
view each (BASELINE.json configs[1]): Hessian-Affine detection + Baumberg, dominant orientation,
RootSIFT description of both images, brute-force FGINN matching, duplicate filtering, LO-RANSAC H.
The batch is pipelined over --workers contexts (host thread + HIP stream each, modsx_match_pairs).
All images are resident in HBM before the timed region.  With --gpus N every rank runs the same
per-GPU workload on its own pairs (pairs shard with no data-path collective: weak scaling).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
INT8_PEAK_TOPS = 5000.0      # dense int8 matrix peak (~2x the 2.5 PF bf16 dense peak)
# profile class -> kernel name as rocprofv3 reports it (default "k_" + class)
KERNEL_OF_CLASS = {"blur_rows": "k_blur_rows_lds", "blur_cols": "k_blur_cols_lds", "match_fginn": "k_match_sweep1"}
# classes whose limiter is vector-ALU issue, with the VALU-busy fraction measured by the SQ counters (profiles/r01_pmc_sq.txt)
VALU_BOUND = {"describe": "0.87", "orientation": "0.70", "baumberg": "0.72", "blur_rows": "0.94", "nms_localize": "0.93"}


def cpu_baseline(rows, cols, seed, budget_s=20.0):
    """The CPU oracle (a restatement of the reference's CPU path, kind 'port') on the same workload,
    single thread, bounded to ~budget_s seconds."""
    import numpy as np
    from mods_amd import synthetic
    from oracle import pyoracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import oracle_pair
    a, b, _ = synthetic.make_pair(rows=rows, cols=cols, nblobs=int(4000 * rows * cols / (768.0 * 1024)), seed=seed)
    n = 0
    t0 = time.time()
    ndesc = 0
    while True:
        r = oracle_pair(O, a, b, seed=1)
        ndesc += len(r["d1"]) + len(r["d2"])
        n += 1
        if time.time() - t0 > budget_s * 0.6 or n >= 8:
            break
    dt = time.time() - t0
    return {"value": n / dt, "unit": "image-pairs/s", "cores": 1, "kind": "port",
            "sample": "%d full pair(s) of the same %dx%d workload in %.1f s, single thread; %d descriptors/pair"
                      % (n, cols, rows, dt, ndesc // n),
            "descriptors_per_s": ndesc / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=768)
    ap.add_argument("--cols", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=64, help="pairs per step per GPU")
    ap.add_argument("--workers", type=int, default=16, help="contexts (thread + stream) per GPU")
    ap.add_argument("--distinct", type=int, default=2, help="distinct synthetic pairs cycled through the batch")
    ap.add_argument("--tilts", type=str, default="", help="e.g. 1,2,3,4,6: synthesise views (configs[2]); default 1 view")
    ap.add_argument("--phi", type=float, default=360.0)
    ap.add_argument("--init-sigma", type=float, default=0.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    import mods_amd
    from mods_amd import synthetic
    seed = 12345 + 1000 * rank
    nblobs = int(4000 * args.rows * args.cols / (768.0 * 1024))
    ctxs = [mods_amd.Context(local_rank) for _ in range(max(1, args.workers))]
    ctx = ctxs[0]
    pairs_host = [synthetic.make_pair(rows=args.rows, cols=args.cols, nblobs=nblobs, seed=seed + 17 * i)
                  for i in range(max(1, args.distinct))]
    H = pairs_host[0][2]
    dev = [(ctx.upload(a), ctx.upload(b)) for a, b, _ in pairs_host]
    imgs1 = [dev[i % len(dev)][0] for i in range(args.batch)]
    imgs2 = [dev[i % len(dev)][1] for i in range(args.batch)]
    params = mods_amd.default_pair_params(ransac_seed=1)

    def barrier():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    views = None
    if args.tilts:
        views = mods_amd.set_vs_pars([1.0], [float(t) for t in args.tilts.split(",")], args.phi, args.init_sigma, 1, [])
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(len(ctxs))

    def run_batch():
        if views is None:
            return mods_amd.match_pairs(ctxs, imgs1, imgs2, params)
        # multi-view pairs: one python thread per context (ctypes releases the GIL inside the library)
        def work(w):
            out = []
            for i in range(w, args.batch, len(ctxs)):
                out.append((i, ctxs[w].match_pair_views(imgs1[i], imgs2[i], views, params)))
            return out
        res = [None] * args.batch
        for part in pool.map(work, range(len(ctxs))):
            for i, r in part:
                res[i] = r
        return res

    for _ in range(args.warmup):
        results = run_batch()
    # The timed region runs WITHOUT the per-launch event brackets (they cost ~5 % of the throughput); per-kernel times of
    # the multi-stream regime come from PROF_STEPS extra, untimed steps right after it.
    barrier()
    t0 = time.perf_counter()
    ndesc = 0
    for _ in range(args.steps):
        results = run_batch()
        for r in results:
            ndesc += r["n_regions"][0] + r["n_regions"][1]
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    res = results[0]
    PROF_STEPS = 2
    for c in ctxs:
        c.profile(True)
    for _ in range(PROF_STEPS):
        run_batch()
    barrier()
    stats = {}
    for c in ctxs:
        for k, v in c.kernel_stats().items():
            d = stats.setdefault(k, dict(ms=0.0, work=0.0, launches=0))
            d["ms"] += v["ms"]; d["work"] += v["work"]; d["launches"] += v["launches"]
        c.profile(False)
    stage = ctx.last_timings()
    # Roofline leg: the timed region runs --workers streams whose kernels time-slice the CUs, so an event pair
    # there brackets queueing as well as execution.  The same pairs are therefore repeated on ONE stream right
    # after the timed region and the per-kernel launch durations are taken from that pass (rank 0 only).
    iso = {}
    if rank == 0:
        ctx.profile(True)
        niso = min(args.batch, 8)
        if views is None:
            # same launch geometry as the timed region: 4 pairs (8 images) per launch set
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
        pairs = world * args.steps * args.batch
        value = pairs / elapsed
        # dominant kernel class by GPU time of the single-stream pass (HIP events on the launch stream)
        name, st = max(iso.items(), key=lambda kv: kv[1]["ms"])
        per_launch_ms = st["ms"] / max(1, st["launches"])
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if views is None and (args.rows, args.cols) == (768, 1024) and os.path.exists(tfile):
            # HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
            # workload (gfx950 correction applied, see profiles/ and DESIGN.md); PMC cannot be sampled in-process
            traffic = json.load(open(tfile)).get(KERNEL_OF_CLASS.get(name, "k_" + name))
        if name == "match_fginn":
            achieved = st["work"] / (st["ms"] * 1e-3) / 1e12 if st["ms"] > 0 else 0.0
            roof = {"kernel": "k_match_fginn", "bound": "mfma", "achieved": achieved, "peak": INT8_PEAK_TOPS,
                    "unit": "TFLOP/s", "frac": achieved / INT8_PEAK_TOPS, "traffic": traffic}
        else:
            achieved = st["work"] / (st["ms"] * 1e-3) / 1e9 if st["ms"] > 0 else 0.0
            roof = {"kernel": KERNEL_OF_CLASS.get(name, "k_" + name), "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic}
        roof["avg_launch_ms"] = per_launch_ms
        roof["algorithmic_work_per_launch"] = st["work"] / max(1, st["launches"])
        tr = stats.get(name)
        roof["timed_region_avg_launch_ms"] = tr["ms"] / max(1, tr["launches"]) if tr else None
        roof["note"] = ("avg_launch_ms: HIP events on the launch stream, single-stream pass run right after the timed "
                        "region (same pairs); timed_region_avg_launch_ms: the same brackets in two untimed steps of the "
                        "multi-stream regime (the timed region itself runs without event brackets), where the kernels "
                        "of %d streams time-slice the CUs" % len(ctxs))
        if name in VALU_BOUND:
            roof["note_bound"] = ("this kernel's limiter is VALU issue (SQ_INSTS_VALU x 4 cycles over SIMD cycles = %s in "
                                  "profiles/, ordered f32/f64 sums per region), not HBM; the schema only offers hbm|mfma, "
                                  "so its HBM fraction is reported as is" % VALU_BOUND[name])
        if name == "patch_sample":
            roof["note_bound"] = ("HBM bytes are this kernel's algorithmic work, but its limiter is the texture addresser: "
                                  "every lane of the 2 x 2 bilinear gathers is its own L1 access (TA busy 55-80 %, "
                                  "2.8 L1 accesses per sample, profiles/ and DESIGN.md section 5)")
        roof["kernels_single_stream_ms_per_pair"] = {k: v["ms"] / min(args.batch, 8) for k, v in iso.items()}
        out = {
            "metric": "image-pairs/sec (1024x768, HessAff+RootSIFT, FGINN match, LO-RANSAC H)",
            "value": value, "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[1]: single %dx%d synthetic pair, 1 view, HessAff+RootSIFT, brute-force "
                                    "FGINN match + LO-RANSAC H" % (args.cols, args.rows)) if views is None else
                                   ("configs[2]: %dx%d synthetic pair, %d affine-synth views (tilts %s, phi %g), "
                                    "HessAff+RootSIFT, MFMA distance matrix + FGINN, LO-RANSAC H"
                                    % (args.cols, args.rows, len(views), args.tilts, args.phi)),
                       "pairs_per_step_per_gpu": args.batch, "workers_per_gpu": len(ctxs),
                       "parallelism": "pairs sharded over ranks, no collective"},
            "descriptors_per_s": ndesc_total / elapsed,
            "descriptors_per_pair": ndesc_total / pairs,
            "result": {"regions": list(res["n_regions"]), "tentatives": res["n_tentatives"], "unique": res["n_unique"],
                       "verified": res["n_verified"], "H_max_abs_err": float(np.abs(res["H"] / res["H"][2, 2] - H).max())},
            "stage_ms_last_pair": stage,
            "kernel_ms_per_pair": {k: v["ms"] / (PROF_STEPS * args.batch) for k, v in stats.items() if v["launches"]},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.rows, args.cols, seed, args.cpu_budget)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    for a_, b_ in dev:
        a_.free(); b_.free()
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

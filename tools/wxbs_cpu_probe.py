#!/usr/bin/env python3
"""CPU seconds per pair of modsx_match_pairs on the WxBS workload, by kind of thread (contexts' own threads / verification helpers),
with H and with F verification: usage wxbs_cpu_probe.py [pairs]"""
import ctypes as C, os, resource, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mods_amd
from mods_amd import synthetic
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if os.environ.get("PROBE_TORCH"):
    import torch
    torch.cuda.set_device(0)
    if os.environ["PROBE_TORCH"] == "2": torch.cuda.synchronize()
imgs = []
ctxs = [mods_amd.Context(0) for _ in range(16)]
for k in range(8):
    a, b, _ = synthetic.make_pair(rows=1080, cols=1920, nblobs=int(4000 * 1920 * 1080 / (1024 * 768)), seed=100 + k)
    imgs.append((ctxs[0].upload(a), ctxs[0].upload(b)))
i1 = [imgs[k % 8][0] for k in range(n)]; i2 = [imgs[k % 8][1] for k in range(n)]
if os.environ.get("PROBE_PROFILE"):
    p0 = mods_amd.default_pair_params(ransac_seed=1, **bench.WXBS)
    for c in ctxs: c.profile(True)
    mods_amd.match_pairs(ctxs, i1[:64], i2[:64], p0)
    for c in ctxs:
        c.synchronize(); c.kernel_stats(); c.profile(False)
if os.environ.get("PROBE_SERIAL"):
    os.environ["MODSX_PAIR_SERIAL"] = "1"; os.environ["MODSX_PAIR_NOSPLIT"] = "1"
for useF in (0, 1, 0, 1):
    p = mods_amd.default_pair_params(ransac_seed=1, useF=useF, **bench.WXBS)
    mods_amd.match_pairs(ctxs, i1[:32], i2[:32], p)
    r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
    mods_amd.match_pairs(ctxs, i1, i2, p)
    dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
    cw, ch = C.c_double(), C.c_double()
    mods_amd.lib().modsx_debug_last_batch_cpu(C.byref(cw), C.byref(ch))
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    print("useF %d: %.1f pairs/s; CPU per pair: process %.4f s = contexts %.4f + helpers %.4f (+ %.4f elsewhere); cores busy %.1f" % (
        useF, n / dt, cpu / n, cw.value / n, ch.value / n, (cpu - cw.value - ch.value) / n, cpu / dt))

"""Executable model (numpy) of the device matcher's DECISION LOGIC (mods_amd/csrc/kernels_match.hip, round 5).

Test infrastructure: `tests/test_match_model_cpu.py` runs it against the oracle's MatchFlannFGINN restatement
(matching/matching.cpp:357-461) on the tie-heavy, clustered and real inputs of the GPU parity tests, so that the part of
the kernels that is logic rather than arithmetic -- which candidates a sweep may drop, which recomputed rows are certain,
how ties fall -- is checked on the CPU, where a GPU is not needed.

What it models
  pack     trains are partitioned by the parity of sum(b) (stable: ascending train index inside a class), each class padded
           to whole stages of 32-row tiles; the sweeps walk "even class, then odd class".  d - |a'|^2 = 2 t + p with p the
           tile's parity, so a key (2 t + p) << 8 | tile code orders by exact distance, and equal distances are always in
           the same class, where slot order is train order.
  sweep 1  per (query, split, lane half) stream: the K smallest GROUP minima (group = the 16 rows of a tile that a lane half
           owns) as (d, tile), ties by ascending tile -- no row index.
  decide   merges the streams into the K smallest groups G[0..K-1] by (d, tile, half), recomputes groups exactly one at a
           time and walks the rows that are CERTAIN: a recomputed row r is certain iff (d_r, tile_r) < (d, tile) of the first
           group not yet recomputed (every row outside the recomputed groups is at or after that group).  The FGINN walk runs
           over the certain rows; when they run out before it ends, the query goes to sweep 2 (which only needs NN0).
  resolve  (queries whose walk outruns the certain rows) by definition: Dmin, nless, nbad, NNj over all trains -- the device
           sweeps again and recomputes the groups below Dmin exactly; ratio >= 1: the query's whole sorted list (k_match_pdf).
"""
import numpy as np

TPS = 4          # tiles per stage of the sweeps: each parity class is padded to whole stages
BIG = np.int64(1) << 40


def row_of(r, hi):
    return 8 * (r >> 2) + 4 * hi + (r & 3)


def pack(d2):
    """virtual slot -> train index (-1 = padding), parity of every virtual tile: the even class, then the odd class, each in train
    order and padded to a multiple of TPS tiles (k_match_pack; on the device the two classes live in two regions of the tile
    array and the sweeps walk them as one virtual sequence)"""
    par = (d2.astype(np.int64).sum(1) & 1).astype(np.int64)
    idx = np.arange(len(d2))
    ev, od = idx[par == 0], idx[par == 1]
    TE = ((len(ev) + 31) // 32 + TPS - 1) // TPS * TPS
    TO = ((len(od) + 31) // 32 + TPS - 1) // TPS * TPS
    perm = -np.ones((TE + TO) * 32, np.int64)
    perm[:len(ev)] = ev
    perm[TE * 32: TE * 32 + len(od)] = od
    tpar = np.zeros(TE + TO, np.int64)
    tpar[TE:] = 1
    return perm, tpar


def ratio_pass(d0, d, sq):
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.float32(d0) / np.float32(d)
    return bool(np.float64(r) <= sq)   # NaN fails


def match_rows(d1, d2, pos2, ratio=0.8, contrad=30.0, nn=50, K=4, S=None, stats=None):
    pdf = ratio >= 1.0      # matching.cpp:397-428: a record per query, closed by the first contradictive neighbour or the last one
    d1 = np.asarray(d1).astype(np.int64)
    d2 = np.asarray(d2).astype(np.int64)
    n1, n2 = len(d1), len(d2)
    sq, cd2 = ratio * ratio, contrad * contrad
    perm, tpar = pack(d2)
    ntiles = len(tpar)
    if S is None:
        S = max(1, min(8, ntiles // 4))
    tps = ((ntiles + S - 1) // S + 3) // 4 * 4
    S = (ntiles + tps - 1) // tps
    D = (d1 * d1).sum(1)[:, None] + (d2 * d2).sum(1)[None, :] - 2 * d1 @ d2.T
    Dp = np.where(perm[None, :] >= 0, D[:, np.maximum(perm, 0)], BIG)          # packed-slot order
    # group minima: group id = tile * 2 + hi; rows of a group in register order
    rows = np.array([[row_of(r, hi) for r in range(16)] for hi in (0, 1)])     # [hi][r]
    gslots = (np.arange(ntiles)[:, None, None] * 32 + rows[None, :, :]).reshape(ntiles * 2, 16)
    Gmin = Dp[:, gslots].min(2)                                               # n1 x (ntiles * 2)
    out = []
    for q in range(n1):
        # sweep 1: K smallest groups of each stream by (d, tile)
        cand = []
        for sp in range(S):
            for hi in (0, 1):
                g = np.arange(sp * tps, min((sp + 1) * tps, ntiles)) * 2 + hi
                if len(g) == 0:
                    continue
                o = np.lexsort((g, Gmin[q, g]))[:K]
                cand += [(int(Gmin[q, g[i]]), int(g[i]) >> 1, int(g[i]) & 1) for i in o if Gmin[q, g[i]] < BIG]
        G = sorted(cand)[:K]                         # (d, tile, hi)
        # decide: certified-prefix walk
        pool = []                                    # recomputed rows (d, slot, tile)
        m = 0
        used = set()
        nbr = []                                     # certified neighbours in order
        res = None                                   # ('accept', j) | ('reject',) | ('sweep2',)
        nrec = 0
        while res is None:
            # next candidate among recomputed, unused rows
            c = min((r for r in pool if r[1] not in used), default=None)
            bound = (G[m][0], G[m][1]) if m < len(G) else ((BIG, 0) if len(G) < K else None)
            # len(G) < K: every group of the problem is in G, so with all of them recomputed everything is certain
            certain = c is not None and (bound is not None and (c[0], c[2]) < bound)
            if certain:
                used.add(c[1])
                nbr.append(c)
                j = len(nbr) - 1
                if j >= 1 and pdf:
                    p0, pj = pos2[perm[nbr[0][1]]], pos2[perm[c[1]]]
                    if j == nn - 1 or ((p0 - pj) ** 2).sum() > cd2:
                        res = ("accept", j)
                elif j >= 1:
                    if ratio_pass(nbr[0][0], c[0], sq):
                        res = ("accept", j)
                    else:
                        p0, pj = pos2[perm[nbr[0][1]]], pos2[perm[c[1]]]
                        if ((p0 - pj) ** 2).sum() > cd2:
                            res = ("reject",)
                        elif j >= nn - 1:
                            res = ("reject",)
                continue
            # cannot certify: recompute the next group, unless only the bound group is left
            usable = len(G) if len(G) < K else K - 1
            if m < usable:
                tile, hi = G[m][1], G[m][2]
                for r in range(16):
                    s = tile * 32 + row_of(r, hi)
                    if perm[s] >= 0:
                        pool.append((int(Dp[q, s]), s, tile))
                m += 1
                nrec += 1
            elif len(G) < K:
                res = ("end",)                       # fewer than nn trains in all: the walk ran off the list
            else:
                res = ("sweep2",)
        if stats is not None:
            stats.append((res[0], nrec))
        row = dict(q=q, t0=-1, t1=-1, tj=-1, nless=0, nbad=0, d0=0.0, d1=0.0, dj=0.0)
        if len(nbr) >= 1:
            row["t0"], row["d0"] = int(perm[nbr[0][1]]), float(nbr[0][0])
        if len(nbr) >= 2:
            row["t1"], row["d1"] = int(perm[nbr[1][1]]), float(nbr[1][0])
        if res[0] == "accept":
            j = res[1]
            row["tj"], row["dj"], row["nless"] = int(perm[nbr[j][1]]), float(nbr[j][0]), j - 1
        elif res[0] == "reject":
            row["nbad"] = 1
        elif res[0] == "sweep2":
            assert len(nbr) >= 2, "NN0 and NN1 must be certain with K >= 4"
            # by definition over all trains (device: sweep 2 + exact recomputation of the event groups)
            t0, d0 = row["t0"], nbr[0][0]
            dm = None
            Dq = D[q]
            order = np.lexsort((np.arange(n2), Dq))
            nless = nbad = 0
            for t in order:
                if t == t0:
                    continue
                if pdf:      # device: k_match_pdf walks the query's whole sorted list
                    if nless + 1 == nn - 1 or ((pos2[t0] - pos2[t]) ** 2).sum() > cd2:
                        row["tj"], row["dj"] = int(t), float(Dq[t])
                        break
                    nless += 1
                    continue
                if ratio_pass(d0, Dq[t], sq):
                    row["tj"], row["dj"] = int(t), float(Dq[t])
                    break
                nless += 1
                if ((pos2[t0] - pos2[t]) ** 2).sum() > cd2:
                    nbad += 1
            row["nless"], row["nbad"] = nless, nbad
        out.append(row)
    return out


def rows_to_tentatives(rows, nn):
    """engine.hip rows_to_tentatives"""
    t = []
    for r in rows:
        if r["t0"] < 0 or r["tj"] < 0 or r["nbad"] != 0 or r["nless"] > nn - 2:
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.float64(np.float32(r["d0"]) / np.float32(r["dj"]))
        t.append((r["q"], r["t0"], r["tj"], r["t1"], r["d0"], r["dj"], r["d1"], np.sqrt(ratio)))
    return t

"""The verification sweep against the reference's compiled degensac (oracle/_ref), as a library: problem generator + runner.

Used by tests/test_ransac_sweep_cpu.py (CPU suite), tests/test_gpu_verify.py (the device rFtH twin) and tools/sweep_ransac_ref.py.
A SWEEP (n, seed0) is the sequence of 4 n problems the tool has always drawn -- per case an H problem of 8-100 tentatives, one of
200-3000, an F problem of 8-100 and one of 150-2400 correspondences (a quarter of them planar scenes, which go through DEGENSAC's
plane-and-parallax branch and the near-degenerate 9 x 9 eigenproblems) -- from ONE sequential generator, so a case is named by
(n is irrelevant) seed0 and its index.  problems() only draws the parameters; run_problem() builds the correspondences, runs
LORANSACFiltering through the reference's degensac and through libmodsx with the same seed and compares trajectory (samples, LO
count), inlier set and kept set.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def problems(n, seed0):
    """The 4 n problems of sweep (n, seed0), in the tool's order: dicts with kind 'H' / 'F', size 'small' / 'mid' and the
    generator arguments.  Draws are sequential: case k of a sweep is the same problem whatever n >= k + 1 is."""
    rng = np.random.default_rng(seed0)
    out = []
    for case in range(n):
        for size in ("small", "mid"):
            T = int(rng.integers(8, 100)) if size == "small" else int(rng.integers(200, 3000))
            frac, noise = float(rng.choice([0.15, 0.3, 0.5, 0.8])), float(rng.choice([0.3, 0.7, 2.0]))
            seed, et = int(rng.integers(1, 1000)), int(rng.integers(0, 3))
            out.append(dict(kind="H", size=size, case=case, seed0=seed0, T=T, frac=frac, noise=noise, gen_seed=seed0 * 100000 + case,
                            seed=seed, et=et))
            if size == "small":
                n_in, n_out = int(rng.integers(8, 60)), int(rng.integers(0, 40))
            else:
                n_in, n_out = int(rng.integers(100, 1200)), int(rng.integers(50, 1200))
            planar, noise = float(rng.choice([0, 0, 0.5, 0.9, 1.0])), float(rng.choice([0.3, 1.0, 2.0]))
            seed, et = int(rng.integers(1, 1000)), int(rng.integers(0, 2))
            out.append(dict(kind="F", size=size, case=case, seed0=seed0, n_in=n_in, n_out=n_out, planar=planar, noise=noise,
                            gen_seed=seed0 * 100000 + case, seed=seed, et=et))
    return out


def name_of(p):
    return "%s %s seed0 %d case %d" % (p["kind"], p["size"], p["seed0"], p["case"])


def run_problem(p, mods=None, oracle=None, device=False):
    """-> dict(same, n_ref, n_here, samples_ref, samples_here, inl_diff, d) for one problem.  `device`: the libmodsx side goes through
    a context (tests/test_gpu_verify.py passes its own callable instead)."""
    from common import synth_corr, synth_two_view
    if mods is None:
        import mods_amd as mods
    if oracle is None:
        import pyoracle as oracle
    if p["kind"] == "H":
        pts, laf, _ = synth_corr(p["T"], p["frac"], noise=p["noise"], seed=p["gen_seed"])
        a = oracle.loransac_h(pts, laf, laf, seed=p["seed"], error_type=p["et"])
        b = mods.loransac_h(pts, laf, laf, seed=p["seed"], error_type=p["et"])
    else:
        pts, laf = synth_two_view(p["gen_seed"], n_in=p["n_in"], n_out=p["n_out"], planar_frac=p["planar"], noise=p["noise"])
        a = oracle.loransac_f(pts, laf, laf, err_threshold=4.0, laf_coef=3.0, seed=p["seed"], error_type=p["et"])
        b = mods.loransac_f(pts, laf, laf, err_threshold=4.0, laf_coef=3.0, seed=p["seed"], error_type=p["et"])
    same = ((a["n"], a["samples"], a["lo_count"]) == (b["n"], b["samples"], b["lo_count"]) and np.array_equal(a["inl"], b["inl"])
            and np.array_equal(a["keep"], b["keep"]))
    d = 0.0
    key = "H" if p["kind"] == "H" else "F"
    if a["n"] and b["n"]:
        Ma, Mb = np.asarray(a[key], float).ravel(), np.asarray(b[key], float).ravel()
        if key == "H":
            if abs(Ma[8]) > 1e-12 and abs(Mb[8]) > 1e-12:
                d = float(np.abs(Ma / Ma[8] - Mb / Mb[8]).max())
        else:
            Ma, Mb = Ma / np.linalg.norm(Ma), Mb / np.linalg.norm(Mb)
            if (Ma * Mb).sum() < 0:
                Mb = -Mb
            d = float(np.abs(Ma - Mb).max())
    return dict(name=name_of(p), same=bool(same), sig_ref=_sig(a), sig_here=_sig(b), n_ref=int(a["n"]), n_here=int(b["n"]), samples_ref=int(a["samples"]),
                samples_here=int(b["samples"]), inl_diff=int(np.count_nonzero(np.asarray(a["inl"]) != np.asarray(b["inl"]))), d=d,
                kind=p["kind"], size=p["size"])


def _sig(r):
    import hashlib
    return hashlib.md5(np.asarray(r["inl"], np.uint8).tobytes() + np.asarray(r["keep"], np.uint8).tobytes()
                       + np.asarray([r["n"], r["samples"], r["lo_count"]], np.int64).tobytes()).hexdigest()[:12]


def reference_signatures(p, tries=6):
    """The reference's result on ONE problem from `tries` fresh processes.  The reference is not reproducible on every problem: its
    least-squares refits go through LAPACK's dsyev_ (scipy's OpenBLAS here), whose kernels depend on where the process' buffers
    happen to lie, and on a near-degenerate problem the last bits of a null vector decide which of two near-tied inlier sets wins
    (e.g. sweep 4 case 1051, F small: 8 kept inliers in two of six processes, none in the other four).  libmodsx's host code is
    reproducible (fixed summation orders, its own eigen-solver).  -> the set of distinct signatures."""
    import json
    import subprocess
    import tempfile
    f = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
    json.dump([p], f); f.close()
    sigs = set()
    try:
        for _ in range(tries):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), f.name], stdout=subprocess.PIPE, check=True,
                                 env=dict(os.environ, OMP_NUM_THREADS="1")).stdout
            sigs.add(json.loads(out.decode().strip().splitlines()[-1])[0]["sig_ref"])
    finally:
        os.unlink(f.name)
    return sigs


def _worker(ps):
    return [run_problem(p) for p in ps]


def run_parallel(ps, procs):
    """The reference's degensac keeps process-global state (HASH_TABLE, the libc PRNG), so the sweep is spread over PROCESSES:
    `python tests/ransac_sweep.py <problems.json>` children (nothing of the parent -- HIP runtime, OpenMP pools -- is inherited),
    each printing its results as one JSON line."""
    if procs <= 1 or len(ps) < 8:
        return _worker(ps)
    import json
    import subprocess
    import tempfile
    chunks = [ps[i::procs] for i in range(procs)]       # interleaved: the mid-size problems are the expensive ones
    files, children = [], []
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for ch in chunks:
        f = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
        json.dump(ch, f); f.close()
        files.append(f.name)
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), f.name], stdout=subprocess.PIPE, env=env))
    by_name = {}
    try:
        for c in children:
            out, _ = c.communicate()
            if c.returncode != 0:
                raise RuntimeError("a sweep child failed (exit code %d)" % c.returncode)
            for r in json.loads(out.decode().strip().splitlines()[-1]):
                by_name[r["name"]] = r
    finally:
        for c in children:
            if c.poll() is None:
                c.kill()
        for f in files:
            os.unlink(f)
    return [by_name[name_of(p)] for p in ps]


if __name__ == "__main__":
    import json
    print(json.dumps(_worker(json.load(open(sys.argv[1])))))

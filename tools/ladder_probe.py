"""Where the time of an MSER-only / HessianAffine-only cviu ladder goes under N contexts (GPU box): pairs/s, host CPU seconds per pair, and the
per-set stage times the library prints with MODSX_HOST_TIMING=1 (stderr), summarised."""
import os, sys, time, threading, itertools, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MODSX_MALLOC_TUNE", "1")
import mods_amd
from mods_amd import synthetic
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import cviu_ladder_steps
det = int(sys.argv[1]) if len(sys.argv) > 1 else 3
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
npairs = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ctxs = [mods_amd.Context(0) for _ in range(workers)]
pairs = [synthetic.make_pair(rows=768, cols=1024, nblobs=4000, seed=12345 + 17 * i) for i in range(8)]
dev = [(ctxs[0].upload(a), ctxs[0].upload(b)) for a, b, _ in pairs]
steps = cviu_ladder_steps(mods_amd, only=(det if det >= 0 else None))   # det < 0: the whole ladder
par = mods_amd.default_pair_params(ransac_seed=1, ori_mrSize=5.1962)
def run():
    nxt = itertools.count(); lock = threading.Lock()
    def work(w):
        while True:
            with lock: i = next(nxt)
            if i >= npairs: return
            ctxs[w].match_ladder(dev[i % 8][0], dev[i % 8][1], steps, par, min_matches=10 ** 6)
    th = [threading.Thread(target=work, args=(w,)) for w in range(workers)]
    [t.start() for t in th]; [t.join() for t in th]
run()
r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
run()
dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
print("det %d workers %d: %.1f pairs/s, host CPU %.3f s user + %.3f s sys per pair (%.1f cores busy)" % (
    det, workers, npairs / dt, (r1.ru_utime - r0.ru_utime) / npairs, (r1.ru_stime - r0.ru_stime) / npairs,
    (r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime) / dt))

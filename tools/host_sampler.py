#!/usr/bin/env python3
"""bench.py (or any script) under the library's host sampling profiler (mods_amd/csrc/hostprof.cpp: ITIMER_PROF + backtrace): where the
host CPU-seconds per pair go.  usage: host_sampler.py <out.txt> <script.py> [script arguments]
Prints per module, per innermost function and per innermost libmodsx function the share of the CPU samples (1 ms of CPU each)."""
import bisect, ctypes, os, runpy, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mods_amd

out, script = sys.argv[1], sys.argv[2]
sys.argv = [script] + sys.argv[3:]
raw = out + ".raw"
# bench.py starts the sampler itself around its first timed region (MODSX_HOST_SAMPLER): SIGPROF during the HIP runtime's start-up
# makes device discovery fail, and the start-up is not what is being asked about
os.environ["MODSX_HOST_SAMPLER"] = raw
try:
    runpy.run_path(script, run_name="__main__")
except SystemExit:
    pass
n = int(open(raw).readline().split()[1])

_syms = {}
def symbols(path):
    if path not in _syms:
        tab = []
        try:
            for fl in ("-n", "-Dn"):
                txt = subprocess.run(["nm", "-C", "--defined-only", fl, path], capture_output=True, text=True).stdout
                for ln in txt.splitlines():
                    p = ln.split(None, 2)
                    if len(p) == 3 and p[1] in "tTwW":
                        tab.append((int(p[0], 16), p[2]))
        except OSError:
            pass
        tab.sort()
        _syms[path] = tab
    return _syms[path]

def name_of(path, off):
    tab = symbols(path)
    i = bisect.bisect_right(tab, (off, "\xff")) - 1
    return tab[i][1][:110] if i >= 0 else "%s+0x%x" % (os.path.basename(path), off)

agg = {"M": {}, "L": {}, "I": {}, "R": {}, "C": {}}
for ln in open(raw):
    p = ln.rstrip("\n").split(" ", 3)
    if p[0] == "M":
        p = ln.rstrip("\n").split(" ", 2)
        agg["M"][os.path.basename(p[2])] = agg["M"].get(os.path.basename(p[2]), 0) + int(p[1])
    elif p[0] == "C":
        p = ln.rstrip("\n").split(" ", 2)
        agg["C"][p[2]] = agg["C"].get(p[2], 0) + int(p[1])
    elif p[0] == "R":
        p = ln.rstrip("\n").split(" ", 3)      # R count path offset <- module
        path, rest = p[2], p[3]
        off, mod = rest.split(" <- ")
        key = ("(no libmodsx frame)" if path.startswith("(outside") else name_of(path, int(off, 16))) + "  <-  " + mod
        agg["R"][key] = agg["R"].get(key, 0) + int(p[1])
    elif p[0] in "LI" and len(p) == 4:
        path, off = p[2], int(p[3], 16)
        key = "(outside libmodsx)" if path.startswith("(outside") else "%s: %s" % (os.path.basename(path), name_of(path, off))
        agg[p[0]][key] = agg[p[0]].get(key, 0) + int(p[1])
with open(out, "w") as f:
    for title, k, top in (("CPU samples per module", "M", 20), ("innermost function", "L", 45), ("innermost libmodsx function on the stack", "I", 60),
                          ("samples inside the runtime / libc: innermost libmodsx function <- module of the innermost frame", "R", 50),
                          ("module chain of the stack, innermost first", "C", 40)):
        f.write("== %s (%d samples of 1 ms CPU)\n" % (title, n))
        for name, c in sorted(agg[k].items(), key=lambda kv: -kv[1])[:top]:
            f.write("%6.2f %%  %7d  %s\n" % (100.0 * c / max(n, 1), c, name))
print(open(out).read())

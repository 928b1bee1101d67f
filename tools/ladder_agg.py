import re, sys, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for l in open(sys.argv[1]):
    m = re.match(r"ladder step (\d+): views ([\d.]+) match ([\d.]+) lists ([\d.]+) verify ([\d.]+) ms", l)
    if m:
        s = int(m.group(1))
        for k, v in zip(("views", "match", "lists", "verify"), m.groups()[1:]):
            tot["step%d %s" % (s, k)] += float(v)
        cnt["step%d" % s] += 1
    m = re.match(r"\s+mser set of (\d+) views: u8 \+ download ([\d.]+) ms, component trees ([\d.]+) ms", l)
    if m:
        tot["mser u8+download"] += float(m.group(2)); tot["mser trees (wall)"] += float(m.group(3)); cnt["mser sets"] += 1
    m = re.match(r"set of (\d+) views: synth ([\d.]+) detect ([\d.]+) orient ([\d.]+) describe ([\d.]+) ms", l)
    if m:
        for k, v in zip(("synth", "detect", "orient", "describe"), m.groups()[1:]):
            tot["set " + k] += float(v)
        cnt["sets"] += 1
n = max(1, cnt["step0"])
print("pairs", n, dict(cnt))
for k in sorted(tot): print("  %-22s %8.2f ms per pair" % (k, tot[k] / n))

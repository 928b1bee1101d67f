R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6zo; mkdir -p $O
cd /tmp && export TMPDIR=/tmp MODSX_PAIR_NOSPLIT=1 MODSX_PAIR_SERIAL=1
for dd in 1 0; do
  rm -rf /tmp/rpg
  AMD_DIRECT_DISPATCH=$dd rocprofv3 --kernel-trace -d /tmp/rpg -o p -- python $R/bench.py --steps 1 --warmup 1 --workers 1 --batch 2 --no-cpu-baseline --no-extra > /tmp/rpg.log 2>&1
  DB=$(find /tmp/rpg -name "*.db" | head -1)
  python $R/tools/rocprof_timeline.py $DB > $O/timeline_dd$dd.txt
  python - <<PY
import re
gaps={}
for l in open("$O/timeline_dd$dd.txt"):
    m=re.match(r"\s*([\d.]+)\s+gap\s+(-?[\d.]+)\s+dur\s+([\d.]+)\s+wgs\s+\d+\s+(\S+)", l)
    if not m: continue
    g=float(m.group(2)); k=re.sub(r"<.*","",m.group(4))
    if g < 50: gaps.setdefault(k,[]).append(g)
print("DD=$dd  median gap before a kernel (gaps < 50 us only):")
for k,v in sorted(gaps.items(), key=lambda kv:-len(kv[1]))[:8]:
    v=sorted(v); print("   %-24s n %5d median %.2f us  p10 %.2f  p90 %.2f" % (k,len(v),v[len(v)//2],v[len(v)//10],v[9*len(v)//10]))
PY
done

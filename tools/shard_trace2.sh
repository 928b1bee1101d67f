#!/bin/bash
# steady-state kernels per pair of the view-sharded path (world 1, one stream): the difference between a long and a short run
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-shardtrace2}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MODSX_PAIR_NOSPLIT=1 MODSX_PAIR_SERIAL=1
for mode in unsharded sharded; do
  a=""; [ $mode = sharded ] && a="--shard views"
  for st in 2 8; do
    rm -rf /tmp/rs_$mode$st
    rocprofv3 --kernel-trace --stats -d /tmp/rs_$mode$st -o p -- python $R/bench.py --steps $st --warmup 1 --workers 1 --batch 4 --no-cpu-baseline --no-extra $a > /tmp/rs.log 2>&1
    DB=$(find /tmp/rs_$mode$st -name "*.db" | head -1)
    python $R/tools/rocprof_summary.py $DB $OUT/k_${mode}_$st.txt "steps $st $a" > /dev/null
  done
done
python - $OUT <<'PY'
import sys
def load(f):
    d={}
    for l in open(f):
        if l.startswith('#') or l.startswith('kernel'): continue
        p=l.split(); d[p[0]]=(int(p[1]),float(p[2]))
    return d
o=sys.argv[1]
for mode in ("unsharded","sharded"):
    a=load("%s/k_%s_2.txt"%(o,mode)); b=load("%s/k_%s_8.txt"%(o,mode))
    print(mode, "per pair (24 pairs between the runs): calls, us")
    tot=0
    for k in sorted(b, key=lambda k:-(b[k][1]-a.get(k,(0,0))[1])):
        dc=(b[k][0]-a.get(k,(0,0))[0])/24; du=(b[k][1]-a.get(k,(0,0))[1])/24; tot+=du
        if dc: print("  %-30s %7.2f %9.1f"%(k,dc,du))
    print("  total us per pair", tot)
PY

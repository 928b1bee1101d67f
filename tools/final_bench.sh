#!/bin/bash
# The measurements DESIGN section 8.1 quotes, in one pass on the GPU box: usage tools/final_bench.sh <dir under gpurun_out>
out=gpurun_out/${1:-final}; mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --config ladder > $out/bench_ladder.json 2> $out/bench_ladder.err
python bench.py --config wxbs > $out/bench_wxbs.json 2> $out/bench_wxbs.err
python bench.py --config views61 --no-cpu-baseline --no-extra > $out/bench_views61.json 2> /dev/null
python bench.py --no-cpu-baseline --no-extra --shard views > $out/rccl_world1.json 2> /dev/null
for W in 1 2 4 8; do python bench.py --no-cpu-baseline --no-extra --loopback $W > $out/loopback_$W.json 2> /dev/null; done
for W in 2 8; do python bench.py --no-cpu-baseline --no-extra --loopback $W --exchange owner > $out/loopback_owner_$W.json 2> /dev/null; done
python tools/latency.py > $out/latency.log 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-24s %8.1f %s" % (f.split("/")[-1], d["value"], d["unit"]))
    except Exception as e:
        print(f, "unreadable:", e)
PY
tail -4 $out/latency.log

#!/bin/bash
# One-GPU legs of the view-sharded path: unsharded reference, RCCL world 1 (fixed ring order / free order), loopback W = 1, 2, 4, 8.
# usage: tools/shard_probe.sh <out dir under gpurun_out> [steps]
out=gpurun_out/${1:-shard}; steps=${2:-6}
mkdir -p $out
B="python bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-extra"
$B > $out/unsharded.json 2> $out/unsharded.err
$B --shard views > $out/rccl_w1.json 2> $out/rccl_w1.err
MODSX_SHARD_FREE_ORDER=1 $B --shard views > $out/rccl_w1_free.json 2> $out/rccl_w1_free.err
for W in 1 2 4 8; do $B --loopback $W > $out/loop_w$W.json 2> $out/loop_w$W.err; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s %8.1f pairs/s  %s" % (f.split("/")[-1], d["value"], json.dumps(d.get("config", {}).get("rccl", d.get("rccl", "")))[:200]))
    except Exception as e:
        print(f, "unreadable:", e)
PY

#!/bin/bash
# Regenerates profiles/ on the GPU box (run through gpurun): kernel-trace summaries (default run + one stream), HBM
# traffic (FETCH_SIZE and WRITE_SIZE in separate --pmc passes) and the SQ counters.  Counter passes use --kernel-trace only.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ONE="--steps 2 --warmup 1 --workers 1 --batch 8 --no-cpu-baseline"
run() { # tag, rocprof args..., -- bench args
  tag=$1; shift
  rm -rf /tmp/rp_$tag
  rocprofv3 "$@" > /tmp/rp_$tag.log 2>&1
  find /tmp/rp_$tag -name "*.db" | head -1
}
DB=$(run def --kernel-trace --stats -d /tmp/rp_def -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline)
python $R/tools/rocprof_summary.py $DB $OUT/r01_kernel_stats_default.txt "python bench.py --steps 5 --warmup 2 --no-cpu-baseline  (default: 16 workers x 64 pairs/step, 4 pairs = 8 images per launch set, 1024x768; kernels of the 16 streams overlap, durations include time-slicing)" > /dev/null
DB=$(run one --kernel-trace --stats -d /tmp/rp_one -o p -- python $R/bench.py --steps 6 --warmup 2 --workers 1 --batch 8 --no-cpu-baseline)
python $R/tools/rocprof_summary.py $DB $OUT/r01_kernel_stats_single_stream.txt "python bench.py --steps 6 --warmup 2 --workers 1 --batch 8 --no-cpu-baseline  (one stream, 4 pairs per launch set)" > /dev/null
DBF=$(run fetch --kernel-trace --pmc FETCH_SIZE -d /tmp/rp_fetch -o p -- python $R/bench.py $ONE)
DBW=$(run write --kernel-trace --pmc WRITE_SIZE -d /tmp/rp_write -o p -- python $R/bench.py $ONE)
python $R/tools/pmc_summary.py $DBF $DBW $OUT/r01_pmc_hbm.txt "python bench.py $ONE (one stream, 4 pairs per launch set)" > /dev/null
DBS=$(run sq --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d /tmp/rp_sq -o p -- python $R/bench.py $ONE)
python $R/tools/pmc_counters.py $DBS $OUT/r01_pmc_sq.txt "python bench.py $ONE (one stream)" > /dev/null
python - <<PY
import json, re
rows = {}
for line in open("$OUT/r01_pmc_hbm.txt"):
    if line.startswith("#") or line.startswith("kernel"): continue
    p = line.rstrip().rsplit(None, 4)
    if len(p) == 5:
        k = re.sub(r"<.*", "", p[0].replace("void ", "").strip())
        rows[k] = max(rows.get(k, 0), int(float(p[3]) + float(p[4])))   # templates: the largest instantiation
json.dump(dict(sorted(rows.items())), open("$OUT/pmc_traffic.json", "w"), indent=1)
PY
ls -la $OUT

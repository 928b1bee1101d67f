#!/bin/bash
# Regenerates the files of one round (ROUND_TAG, default r03) of profiles/ on the GPU box (run through gpurun; output lands in gpurun_out/profiles, copy
# what is to be judged into profiles/).  Kernel-trace summaries and counter passes are separate rocprofv3 runs; counter
# passes use --kernel-trace only.
R=$GRAFT_REPO_ROOT
TAG=${ROUND_TAG:-r05}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# one context, one stream, whole launch sets: the lone-pair helpers (an image in three parts, image 2 on a peer context) are switched off
export MODSX_PAIR_NOSPLIT=1 MODSX_PAIR_SERIAL=1
ONE="--steps 2 --warmup 1 --workers 1 --batch 4 --no-cpu-baseline --no-extra"
ONE1="--config views1 --steps 2 --warmup 1 --workers 1 --batch 8 --no-cpu-baseline --no-extra"
run() { # tag, rocprof args..., -- command
  tag=$1; shift
  rm -rf /tmp/rp_$tag
  rocprofv3 "$@" > /tmp/rp_$tag.log 2>&1
  find /tmp/rp_$tag -name "*.db" | head -1
}
# 1. the default bench (31 views, 16 streams): durations include time-slicing between the streams
DB=$(run def --kernel-trace --stats -d /tmp/rp_def -o p -- python $R/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2)
python $R/tools/rocprof_summary.py $DB $OUT/${TAG}_kernel_stats_default.txt "python bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2  (default workload: 31 views per image, 16 workers x 16 pairs/step; kernels of the 16 streams overlap, durations include time-slicing)" > /dev/null 2> $OUT/${TAG}_default.err || { echo "default summary failed: DB=[$DB]"; tail -5 /tmp/rp_def.log; ls -la /tmp/rp_def | head; } > $OUT/${TAG}_default.diag 2>&1
# 2. one stream, same workload: the durations the roofline objects of bench.py are compared with
DB=$(run one --kernel-trace --stats -d /tmp/rp_one -o p -- python $R/bench.py $ONE)
python $R/tools/rocprof_summary.py $DB $OUT/${TAG}_kernel_stats_single_stream.txt "python bench.py $ONE  (31 views per image, one stream)" > /dev/null
DB=$(run one1 --kernel-trace --stats -d /tmp/rp_one1 -o p -- python $R/bench.py $ONE1)
python $R/tools/rocprof_summary.py $DB $OUT/${TAG}_kernel_stats_views1_single_stream.txt "python bench.py $ONE1  (configs[1]: 1 view, one stream, 4 pairs per launch set -- comparable with r01_kernel_stats_single_stream.txt)" > /dev/null
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
DBF=$(run fetch --kernel-trace --pmc FETCH_SIZE -d /tmp/rp_fetch -o p -- python $R/bench.py $ONE)
DBW=$(run write --kernel-trace --pmc WRITE_SIZE -d /tmp/rp_write -o p -- python $R/bench.py $ONE)
python $R/tools/pmc_summary.py $DBF $DBW $OUT/${TAG}_pmc_hbm.txt "python bench.py $ONE (31 views, one stream)" > /dev/null
DBF=$(run fetch1 --kernel-trace --pmc FETCH_SIZE -d /tmp/rp_fetch1 -o p -- python $R/bench.py $ONE1)
DBW=$(run write1 --kernel-trace --pmc WRITE_SIZE -d /tmp/rp_write1 -o p -- python $R/bench.py $ONE1)
python $R/tools/pmc_summary.py $DBF $DBW $OUT/${TAG}_pmc_hbm_views1.txt "python bench.py $ONE1 (configs[1], one stream, 4 pairs per launch set: comparable with r01_pmc_hbm.txt)" > /dev/null
# 4. SQ counters
DBS=$(run sq --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/rp_sq -o p -- python $R/bench.py $ONE)
python $R/tools/pmc_counters.py $DBS $OUT/${TAG}_pmc_sq.txt "python bench.py $ONE (31 views, one stream); GRBM_GUI_ACTIVE is summed over the 8 XCDs" > /dev/null
# 5. the matcher alone on real descriptors: 8 views (10 k x 10 k), 8 views doubled (20 k x 20 k), 31 views (24 k x 24 k), doubled (48 k)
for spec in "m10k:" "m20k:--rep 2" "m24k:--tilts 1,2,4,6,8 --phi 120" "m48k:--tilts 1,2,4,6,8 --phi 120 --rep 2"; do
  tag=${spec%%:*}; a=${spec#*:}
  DB=$(run $tag --kernel-trace --stats -d /tmp/rp_$tag -o p -- python $R/tools/bench_match.py $a --reps 10)
  grep -h "N=" /tmp/rp_$tag.log | tail -1 > /tmp/rp_$tag.line
  python $R/tools/rocprof_summary.py $DB $OUT/${TAG}_match_$tag.txt "python tools/bench_match.py $a --reps 10   $(cut -c1-220 /tmp/rp_$tag.line)" > /dev/null
done
DBM=$(run msq --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/rp_msq -o p -- python $R/tools/bench_match.py --tilts 1,2,4,6,8 --phi 120 --reps 3)
python $R/tools/pmc_counters.py $DBM $OUT/${TAG}_match_pmc_sq.txt "python tools/bench_match.py --tilts 1,2,4,6,8 --phi 120 --reps 3 (24 k x 24 k real descriptors)" > /dev/null
DBF=$(run mf --kernel-trace --pmc FETCH_SIZE -d /tmp/rp_mf -o p -- python $R/tools/bench_match.py --tilts 1,2,4,6,8 --phi 120 --reps 3)
DBW=$(run mw --kernel-trace --pmc WRITE_SIZE -d /tmp/rp_mw -o p -- python $R/tools/bench_match.py --tilts 1,2,4,6,8 --phi 120 --reps 3)
python $R/tools/pmc_summary.py $DBF $DBW $OUT/${TAG}_match_pmc_hbm.txt "python tools/bench_match.py --tilts 1,2,4,6,8 --phi 120 --reps 3 (24 k x 24 k real descriptors)" > /dev/null
python - <<PY
import json, re
def table(path):
    rows = {}
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel"): continue
        p = line.rstrip().rsplit(None, 4)
        if len(p) == 5:
            k = re.sub(r"<.*", "", p[0].replace("void ", "").strip())
            rows[k] = max(rows.get(k, 0), int(float(p[3]) + float(p[4])))   # fetch x2 + write; templates: the largest instantiation
    return rows
t = table("$OUT/${TAG}_pmc_hbm.txt")
m = table("$OUT/${TAG}_match_pmc_hbm.txt")
t["k_match_total_views31"] = sum(v for k, v in m.items() if k.startswith("k_match_"))
hdr = open("/tmp/rp_m24k.line").read()
mm = re.search(r"N=(\d+) M=(\d+)", hdr)
try:   # the describe chunk of the profiled run: its algorithmic bytes (bench.py's roofline_describe), so that a later run can scale the traffic to its own chunk
    line = [l for l in open("/tmp/rp_fetch.log") if l.startswith("{")][-1]
    t["describe_chunk_algorithmic_bytes"] = json.loads(line)["roofline_describe"]["algorithmic_work_per_launch"]
except Exception as e:
    print("no roofline_describe in the fetch pass:", e)
if mm:
    t["k_match_compulsory_bytes"] = (int(mm.group(1)) + int(mm.group(2))) * 128 + int(mm.group(1)) * 32
try:   # HBM bytes of the whole pipeline per pair: every kernel's FETCH_SIZE x2 + WRITE_SIZE of the one-stream pass over its pairs
    tot = 0
    for ln in open("$OUT/${TAG}_pmc_hbm.txt"):
        if ln.startswith("#") or ln.startswith("kernel"): continue
        p = ln.rstrip().rsplit(None, 4)
        if len(p) == 5: tot += float(p[1]) * (float(p[3]) + float(p[4])) if p[1].replace(".", "").isdigit() else 0
    npairs = 0          # pairs of bench.py $ONE = its k_match_sweep1 launches (one matching problem per pair)
    for ln in open("$OUT/${TAG}_kernel_stats_single_stream.txt"):
        if ln.startswith("k_match_sweep1"): npairs += int(ln.split()[-10])
    t["pairs_in_one_stream_profile"] = npairs
    t["hbm_counter_GB_per_pair"] = tot / npairs / 1e9
except Exception as e:
    print("no per-pair total:", e)
t["k_match_problem"] = ("%s x %s real 31-view descriptors (tools/bench_match.py --tilts 1,2,4,6,8 --phi 120)" % (mm.group(1), mm.group(2))) if mm else "24 k x 24 k real 31-view descriptors"
json.dump(dict(sorted(t.items())), open("$OUT/pmc_traffic_${TAG}.json", "w"), indent=1)
PY
ls -la $OUT

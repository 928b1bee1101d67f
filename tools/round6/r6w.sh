R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6w; mkdir -p $O
cd $R
timeout 600 python bench.py --config ladder --no-cpu-baseline > $O/ladder.json 2> $O/ladder.err
python - <<PY
import json
d=json.loads([l for l in open("$O/ladder.json") if l.startswith("{")][-1])
print("ladder", round(d["value"],1), d.get("value_with_step_barrier",{}).get("value"), d.get("host_cpu_s_per_pair_rank0"), d.get("ladder_parts"))
PY
MODSX_HOST_WAIT=flag timeout 600 python bench.py --config ladder --no-cpu-baseline --no-extra > $O/ladder_flag.json 2> $O/ladder_flag.err
python - <<PY
import json
d=json.loads([l for l in open("$O/ladder_flag.json") if l.startswith("{")][-1])
print("ladder flag", round(d["value"],1), d.get("host_cpu_s_per_pair_rank0"))
PY
python tools/mser_check.py 5 3 | tail -1

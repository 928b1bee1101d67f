R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6y; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log); tail -3 $O/pytest.log
timeout 600 python bench.py --config ladder --no-cpu-baseline > $O/ladder.json 2> $O/ladder.err
python tools/bench_line.py ladder < $O/ladder.json
python - <<PY
import json
d=json.loads([l for l in open("$O/ladder.json") if l.startswith("{")][-1])
print(d.get("ladder_parts"))
PY

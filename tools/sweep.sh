#!/bin/bash
# throughput sweep over (workers, batch); prints workers batch pairs/s
for wb in "16 64" "16 128" "8 64" "12 96" "24 96"; do
  set -- $wb
  timeout 300 python bench.py --no-cpu-baseline --workers $1 --batch $2 --steps 8 2>/dev/null > /tmp/o.json
  python - <<'PY'
import json
d = json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print(d["config"]["workers_per_gpu"], d["config"]["pairs_per_step_per_gpu"], round(d["value"], 1))
PY
done

#!/bin/bash
for cfg in "0 8 32" "0 12 48" "0 16 64" "0 20 80" "1 8 32" "1 12 48"; do
  set -- $cfg
  v=$(MODSX_MATCH_BATCH=$1 python bench.py --no-cpu-baseline --workers $2 --batch $3 --steps 12 2>/dev/null | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["value"],1))')
  echo "batch=$1 workers=$2 pairs/step=$3: $v"
done

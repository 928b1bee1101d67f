#!/bin/bash
python bench.py --no-cpu-baseline --steps 200 --warmup 3 > /tmp/bu.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do rocm-smi --showuse 2>/dev/null | grep -i "GPU use" | awk '{print $NF}' | tr '\n' ' '; sleep 0.5; done
echo
tail -1 /tmp/bu.json | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])'

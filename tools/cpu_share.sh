#!/bin/bash
# throughput when the process may only use a fraction of the host cores (what a rank gets on a full 8-GPU node)
nproc
run() { python bench.py --no-cpu-baseline --steps 15 2>/dev/null | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["value"],1))'; }
echo "all cores: $(run)"
for n in 32 16 8; do echo "taskset $n cores: $(taskset -c 0-$((n-1)) python bench.py --no-cpu-baseline --steps 15 2>/dev/null | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["value"],1))')"; done

#!/usr/bin/env python3
"""prints value / value_with_step_barrier of a bench.py JSON line read from stdin (tag = argv[1])"""
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
b = j.get("value_with_step_barrier") or {}
print(sys.argv[1] if len(sys.argv) > 1 else "", round(j["value"], 1), round(b.get("value", 0), 1), round(j.get("host_cpu_s_per_pair_rank0", 0), 4))

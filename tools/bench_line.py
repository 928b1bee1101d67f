#!/usr/bin/env python3
"""prints value / value_with_step_barrier / host CPU per pair (total: context threads + other threads, system time) of a bench.py JSON line read from stdin (tag = argv[1])"""
import json, sys
lines = [l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")]
j = json.loads(lines[-1])
b = j.get("value_with_step_barrier") or {}
sp = j.get("host_cpu_split_s_per_pair") or {}
print(sys.argv[1] if len(sys.argv) > 1 else "", round(j["value"], 1), round(b.get("value", 0), 1), round(j.get("host_cpu_s_per_pair_rank0", 0), 4),
      "ctx %.4f other %.4f sys %.4f" % (sp.get("context_threads", 0), sp.get("other_threads", 0), sp.get("of_which_system_time", 0)))

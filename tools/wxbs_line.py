#!/usr/bin/env python3
"""prints the H / F figures of a `bench.py --config wxbs` JSON line read from stdin (tag = argv[1])"""
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
w = j["wxbs"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "H", round(w["H"]["pairs_per_s"], 1), "F", round(w["F"]["pairs_per_s"], 1), "(per-step calls %.1f)" % w["F"].get("pairs_per_s_one_call_per_step", 0),
      "verify ms/pair", round(w["F"]["verify_ms_per_pair"], 2), "rfth ms/loop", round(w["F"]["rfth_ms_per_loop"], 2),
      "cpu s/pair H %.4f F %.4f cores busy F %.1f sys %.2f" % (j.get("host_cpu_s_per_pair_rank0", 0), w["F"].get("host_cpu_s_per_pair", 0), w["F"].get("host_cores_busy", 0), w["F"].get("host_cpu_system_share", 0)),
      "cpu/pair: contexts %.4f helpers %.4f" % (w["F"].get("cpu_s_per_pair_context_threads", 0), w["F"].get("cpu_s_per_pair_verification_helpers", 0)),
      "batches", w["F"]["rfth_device_batches"], "host share", round(w["F"]["host_ransac_share_of_wall"], 3), {k: round(v, 2) for k, v in w["F"].get("rfth_ms_per_loop_parts", {}).items()})

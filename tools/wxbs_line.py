#!/usr/bin/env python3
"""prints the H / F figures of a `bench.py --config wxbs` JSON line read from stdin (tag = argv[1])"""
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
w = j["wxbs"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "H", round(w["H"]["pairs_per_s"], 1), "F", round(w["F"]["pairs_per_s"], 1),
      "verify ms/pair", round(w["F"]["verify_ms_per_pair"], 2), "rfth ms/loop", round(w["F"]["rfth_ms_per_loop"], 2),
      "batches", w["F"]["rfth_device_batches"], "host share", round(w["F"]["host_ransac_share_of_wall"], 3), {k: round(v, 2) for k, v in w["F"].get("rfth_ms_per_loop_parts", {}).items()})

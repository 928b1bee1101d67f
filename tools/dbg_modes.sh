#!/bin/bash
for m in 0 1 2 3 5 6; do
  MODSX_DBG=$m bash tools/prof_single.sh
  echo "mode $m: $(grep k_orientation gpurun_out/prof_single.txt)"
  echo "mode $m: $(grep k_baumberg gpurun_out/prof_single.txt)"
done

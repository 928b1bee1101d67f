#!/bin/bash
for m in 0 1 2 3 4; do
  MODSX_DBG=$m bash tools/prof_single.sh
  echo "mode $m: $(grep k_describe gpurun_out/prof_single.txt)"
done

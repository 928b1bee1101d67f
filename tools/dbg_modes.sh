#!/bin/bash
for m in 0 1; do
  MODSX_DBG=$m bash tools/prof_single.sh
  echo "mode $m: $(grep k_nms_localize gpurun_out/prof_single.txt)"
done

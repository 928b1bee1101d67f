#!/bin/bash
cp mods_amd/libmodsx.so /tmp/new.so
run() { python bench.py --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["value"],1))'; }
for rep in 1 2 3; do
for v in old new; do
  if [ $v = old ]; then cp mods_amd/libmodsx_old.so mods_amd/libmodsx.so; else cp /tmp/new.so mods_amd/libmodsx.so; fi
  for q in default 8; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    echo "$v queues=$q: $(run)"
  done
done
done
cp /tmp/new.so mods_amd/libmodsx.so

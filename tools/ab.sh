#!/bin/bash
# A/B two builds on the same box: mods_amd/libmodsx_old.so vs mods_amd/libmodsx.so
cp mods_amd/libmodsx.so /tmp/new.so
for rep in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp mods_amd/libmodsx_old.so mods_amd/libmodsx.so; else cp /tmp/new.so mods_amd/libmodsx.so; fi
    echo "$v $(python bench.py --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])')"
  done
done
cp /tmp/new.so mods_amd/libmodsx.so

# usage: ab_bench.sh out.log lib1 lib2 ...  (variants built by tools/build_variant.sh; "main" = mods_amd/libmodsx.so)   (default bench, 5 steps, alternating)
out=$1; shift
mkdir -p $(dirname $out)
for rep in 1 2; do
for v in "$@"; do
  lib=mods_amd/libmodsx_$v.so; [ "$v" = "main" ] && lib=mods_amd/libmodsx.so
  val=$(MODSX_LIB=$lib timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value'],1))")
  echo "$v $val" >> $out
done
done

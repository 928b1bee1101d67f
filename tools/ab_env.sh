# usage: ab_env.sh out.log "ENV1=.. ENV2=.." "..." ...   each argument is an environment prefix ("-" = none); default bench, 5 steps, two alternating rounds
out=$1; shift
mkdir -p $(dirname $out)
for rep in 1 2; do
for v in "$@"; do
  e="$v"; [ "$v" = "-" ] && e=""
  val=$(env $e timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['value'],1), round(j.get('kernels_single_stream_ms_per_pair',{}).get('nms_localize',0),4))")
  echo "[$v] $val" >> $out
done
done

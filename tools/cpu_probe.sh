# usage: cpu_probe.sh "<env>" <bench args...>: pairs/s + host CPU seconds per pair + cgroup throttling of one bench run
e="$1"; shift; [ "$e" = "-" ] && e="MODSX_NOOP=1"
s0=$(grep -E "usage_usec|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | awk '{print $2}' | tr '\n' ' ')
out=$(env $e python bench.py --no-extra --no-cpu-baseline "$@" 2>/dev/null | tail -1)
s1=$(grep -E "usage_usec|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | awk '{print $2}' | tr '\n' ' ')
python - "$e" "$s0" "$s1" "$out" "$*" <<'PY'
import sys, json
e, s0, s1, out, args = sys.argv[1:6]
a = list(map(int, s0.split())); b = list(map(int, s1.split()))
j = json.loads(out)
print("[%s] %s: %.1f pairs/s; whole run: %.1f CPU-s, throttled %d periods / %.1f thread-s" % (e, args, j["value"], (b[0]-a[0])/1e6, b[1]-a[1], (b[2]-a[2])/1e6))
PY

R=$GRAFT_REPO_ROOT; cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
for i in 1 2 3; do
MODSX_POOL_HELP=0 timeout 600 python bench.py --config ladder --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py ladder_nohelp
timeout 600 python bench.py --config ladder --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py ladder_help
done

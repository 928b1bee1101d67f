R=$GRAFT_REPO_ROOT; cd $R
cat /sys/kernel/mm/transparent_hugepage/enabled
for i in 1 2; do
MODSX_MSER_NOHUGE=1 python tools/mser_check.py 2 3 | tail -1 | sed 's/^/nohuge /'
python tools/mser_check.py 2 3 | tail -1 | sed 's/^/huge   /'
done
for pf in 6 8 16 24; do MODSX_MSER_PF=$pf python tools/mser_check.py 2 3 | tail -1 | sed "s/^/pf$pf /"; done
for i in 1 2; do
MODSX_MSER_NOHUGE=1 timeout 600 python bench.py --config ladder --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py ladder_nohuge
timeout 600 python bench.py --config ladder --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py ladder_huge
done

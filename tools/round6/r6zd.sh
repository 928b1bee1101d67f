R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6zd; mkdir -p $O
cd $R
export MODSX_BENCH_NO_UPLOAD_LEG=1
for v in base b1 b2w4 b2w3; do
  if [ "$v" = base ]; then L=$R/mods_amd/libmodsx.so; else L=$R/mods_amd/libmodsx_$v.so; fi
  export MODSX_LIB=$L
  (cd /tmp && export TMPDIR=/tmp MODSX_PAIR_NOSPLIT=1 MODSX_PAIR_SERIAL=1 && rm -rf /tmp/rp1 && rocprofv3 --kernel-trace --stats -d /tmp/rp1 -o p -- python $R/bench.py --steps 2 --warmup 1 --workers 1 --batch 4 --no-cpu-baseline --no-extra > /tmp/rp1.log 2>&1; python $R/tools/rocprof_summary.py $(find /tmp/rp1 -name "*.db" | head -1) $O/kernels_$v.txt "one stream $v" | grep -E "baumberg" | head -2 | sed "s/^/$v /")
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py $v
done
for v in base b2w4 b2w3; do
  if [ "$v" = base ]; then L=$R/mods_amd/libmodsx.so; else L=$R/mods_amd/libmodsx_$v.so; fi
  MODSX_LIB=$L timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python tools/bench_line.py $v
done
MODSX_LIB=$R/mods_amd/libmodsx_b2w3.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "affine or baumberg or pair or cat" 2>&1 | grep -E "passed|failed"

R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b; mkdir -p $O
for v in base kog koc kob kov bk3 bk4; do
  if [ "$v" = base ]; then L=$R/mods_amd/libmodsx.so; else L=$R/mods_amd/libmodsx_$v.so; fi
  MODSX_LIB=$L bash $R/tools/prof_cmd.sh k_$v "k_describe\|k_baumberg\|k_orientation" python $R/tools/bench_detect.py --desc 1 --reps 5 >> $O/prof.txt 2>&1
done
for v in bk3 bk4; do
  MODSX_LIB=$R/mods_amd/libmodsx_$v.so timeout 900 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "affine or baumberg" > $O/pytest_$v.txt 2>&1
done
cd $R && bash tools/ab_bench.sh base bk3 bk4 kog koc > $O/ab.txt 2>&1
cat $O/prof.txt $O/ab.txt; for f in $O/pytest_*.txt; do tail -n 3 $f; done

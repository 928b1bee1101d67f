R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_new.txt 2>&1
MODSX_LIB=$R/mods_amd/libmodsx_ordonly.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "descri or pair or sift" > $O/pytest_ordonly.txt 2>&1
for v in base r5 ordonly; do
  if [ "$v" = base ]; then L=$R/mods_amd/libmodsx.so; else L=$R/mods_amd/libmodsx_$v.so; fi
  MODSX_LIB=$L bash $R/tools/prof_cmd.sh c_$v "k_describe" python $R/tools/bench_detect.py --desc 1 --reps 5 >> $O/prof.txt 2>&1
done
bash tools/ab_bench.sh base r5 > $O/ab.txt 2>&1
cat $O/prof.txt $O/ab.txt; for f in $O/pytest_*.txt; do tail -n 3 $f; done

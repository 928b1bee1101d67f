R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6h; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_views.py -x -q -m gpu > $O/pytest.txt 2>&1
for v in base dr4 dr3; do
  if [ "$v" = base ]; then L=$R/mods_amd/libmodsx.so; else L=$R/mods_amd/libmodsx_$v.so; fi
  MODSX_LIB=$L bash $R/tools/prof_cmd.sh h_$v "k_describe\|k_blur_hess" python $R/tools/bench_detect.py --desc 1 --reps 5 >> $O/prof.txt 2>&1
done
MODSX_LIB=$R/mods_amd/libmodsx_dr4.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "descri or pair or sift" > $O/pytest_dr4.txt 2>&1
bash tools/ab_bench.sh base dr4 dr3 > $O/ab.txt 2>&1
grep -n "passed\|failed" $O/pytest.txt $O/pytest_dr4.txt; cat $O/prof.txt $O/ab.txt | cut -c1-200

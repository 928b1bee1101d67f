R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6a; mkdir -p $O
$R/tools/ubench/wave_simd > $O/wave_simd.txt 2>&1
for v in base dw1 dw2 dw3; do
  if [ "$v" = base ]; then L=$R/mods_amd/libmodsx.so; else L=$R/mods_amd/libmodsx_$v.so; fi
  MODSX_LIB=$L bash $R/tools/prof_cmd.sh d_$v "k_describe\|k_sample_rows\|k_blur_cols" python $R/tools/bench_detect.py --desc 1 --reps 5 >> $O/prof.txt 2>&1
done
for v in dw1 dw3; do
  MODSX_LIB=$R/mods_amd/libmodsx_$v.so timeout 900 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "descri" > $O/pytest_$v.txt 2>&1
done
cd $R && bash tools/ab_bench.sh base dw1 dw2 dw3 > $O/ab.txt 2>&1
cat $O/wave_simd.txt $O/prof.txt $O/ab.txt; tail -3 $O/pytest_*.txt

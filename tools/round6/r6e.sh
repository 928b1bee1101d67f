R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
tools/ubench/int_rates > $O/int_rates.txt 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
for v in r6base new; do
  if [ "$v" = new ]; then L=$R/mods_amd/libmodsx.so; else L=$R/mods_amd/libmodsx_$v.so; fi
  MODSX_LIB=$L bash $R/tools/prof_cmd.sh e_$v "k_describe\|k_baumberg\|k_orientation\|k_sample_rows\|k_blur_cols" python $R/tools/bench_detect.py --desc 1 --reps 5 >> $O/prof.txt 2>&1
  MODSX_LIB=$L bash $R/tools/pmc_cmd.sh e_$v "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "k_describe\|k_baumberg\|k_orientation\|k_sample_rows\|k_blur_cols" python $R/tools/bench_detect.py --desc 1 --reps 3 >> $O/pmc.txt 2>&1
done
sed -i 's/^base /r6base /' /dev/null
bash tools/ab_bench.sh r6base base > $O/ab.txt 2>&1
cat $O/int_rates.txt; grep -n "passed\|failed" $O/pytest.txt; cat $O/prof.txt $O/pmc.txt $O/ab.txt

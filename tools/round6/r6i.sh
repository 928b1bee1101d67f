R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6i; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_views.py -x -q -m gpu > $O/pytest.txt 2>&1
for v in r6i0 base; do
  if [ "$v" = base ]; then L=$R/mods_amd/libmodsx.so; else L=$R/mods_amd/libmodsx_$v.so; fi
  MODSX_LIB=$L bash $R/tools/prof_cmd.sh i_$v "k_orientation" python $R/tools/bench_detect.py --desc 1 --reps 5 >> $O/prof.txt 2>&1
  MODSX_LIB=$L bash $R/tools/pmc_cmd.sh i_$v "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "k_orientation" python $R/tools/bench_detect.py --desc 1 --reps 3 >> $O/pmc.txt 2>&1
done
bash tools/ab_bench.sh r6i0 base > $O/ab.txt 2>&1
MODSX_BENCH_NO_UPLOAD_LEG=1 python tools/host_sampler.py $O/host_profile.txt bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/host_sampler.log 2>&1
grep -n "passed\|failed" $O/pytest.txt; cat $O/prof.txt $O/pmc.txt $O/ab.txt | cut -c1-200; head -100 $O/host_profile.txt | cut -c1-200

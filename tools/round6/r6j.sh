R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "affine or baumberg or pair or keypoint" > $O/pytest.txt 2>&1
MODSX_LIB=$R/mods_amd/libmodsx_bk4.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "affine or baumberg or pair or keypoint" > $O/pytest_bk4.txt 2>&1
for v in base bk4; do
  if [ "$v" = base ]; then L=$R/mods_amd/libmodsx.so; else L=$R/mods_amd/libmodsx_$v.so; fi
  MODSX_LIB=$L bash $R/tools/prof_cmd.sh j_$v "k_baumberg" python $R/tools/bench_detect.py --desc 1 --reps 5 >> $O/prof.txt 2>&1
  MODSX_LIB=$L bash $R/tools/pmc_cmd.sh j_$v "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "k_baumberg" python $R/tools/bench_detect.py --desc 1 --reps 3 >> $O/pmc.txt 2>&1
done
bash tools/ab_bench.sh base bk4 > $O/ab.txt 2>&1
MODSX_BENCH_NO_UPLOAD_LEG=1 timeout 600 python tools/host_sampler.py $O/host_profile.txt bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/host_sampler.log 2>&1
grep -n "passed\|failed" $O/pytest.txt $O/pytest_bk4.txt; cat $O/prof.txt $O/pmc.txt $O/ab.txt | cut -c1-200; head -140 $O/host_profile.txt | cut -c1-200

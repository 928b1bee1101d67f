#!/bin/bash
# TEST INFRASTRUCTURE (build container only: needs /root/reference).  Builds two tracing variants for diffing the trajectory
# of exp_ransacFcustom call by call (how the slcm term order and the 4-point u2h branch were found in round 4):
#   $OUT/oracle/_ref/libdegensac_ref.so   the reference's degensac with the call sites of exp_ranF.c renamed (-Dname=tr_name)
#                                         onto trace_shim.c wrappers that print and forward -- no reference file is modified
#   $OUT/libmodsx_trace.so                libmodsx with ransac_f.cpp compiled -DMODSX_TRACE_RANSAC (same print format)
# Use: PYTHONPATH=$OUT python -c "from oracle import pyoracle as O; O.loransac_f(...)" 2> ref.log
#      MODSX_LIB=$OUT/libmodsx_trace.so python -c "import mods_amd; mods_amd.loransac_f(...)" 2> mine.log; diff ref.log mine.log
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
REF=${REF:-/root/reference}; OUT=${OUT:-/tmp/trace}; O=$ROOT/oracle
mkdir -p $OUT/obj $OUT/oracle/_ref
SCIPY_LIBS=$(python3 -c "import scipy,os;print(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)),'scipy.libs'))")
OPENBLAS=$(ls $SCIPY_LIBS/libscipy_openblas-*.so | head -1)
REFFLAGS="-O2 -w -fcommon -fPIC -ffp-contract=off -I$REF -I$REF/degensac -Ddsyev_=scipy_dsyev_ -Ddgesvd_=scipy_dgesvd_ -Ddgeqp3_=scipy_dgeqp3_"
DEG="DegUtils exp_ranF exp_ranH Ftools hash Htools ranF ranH2el ranH rtools utools lapwrap"
for f in $DEG; do
  extra=""; case $f in exp_ranH|exp_ranF) extra="-Dtime=modsx_ref_time";; esac
  case $f in Ftools|Htools|exp_ranH|lapwrap|ranH2el) extra="$extra -include $O/ref_lp64.h";; esac
  case $f in exp_ranF) extra="$extra -Dchecksample=tr_checksample -DrFtH=tr_rFtH -DinnerH=tr_innerH -Dnsamples=tr_nsamples -Dinlidxs=tr_inlidxs -Dnullspace=tr_nullspace -Drroots3=tr_rroots3 -Dall_ori_valid=tr_all_ori_valid";; esac
  gcc $REFFLAGS $extra -c $REF/degensac/$f.c -o $OUT/obj/$f.o; done
for f in $REF/matutls/*.c; do b=$(basename $f .c); [ $b = svd2 ] && continue; gcc -O2 -w -fPIC -ffp-contract=off -c $f -o $OUT/obj/mu_$b.o; done
rm -f $OUT/obj/libmatutls.a; ar rcs $OUT/obj/libmatutls.a $OUT/obj/mu_*.o
gcc -O2 -w -fPIC -c $O/ref_shim.c -o $OUT/obj/ref_shim.o; gcc -O2 -w -fPIC -c $HERE/trace_shim.c -o $OUT/obj/trace_shim.o
objs=""; for f in $DEG; do objs="$objs $OUT/obj/$f.o"; done
gcc -shared -o $OUT/oracle/_ref/libdegensac_ref.so $objs $OUT/obj/ref_shim.o $OUT/obj/trace_shim.o $OUT/obj/libmatutls.a $OPENBLAS -Wl,-rpath,$SCIPY_LIBS -lm
cp $O/liboracle.so $O/pyoracle.py $OUT/oracle/; touch $OUT/oracle/__init__.py
cd $ROOT/mods_amd/csrc && make -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -Xarch_host -mavx2 -DMODSX_TRACE_RANSAC -x hip -c ransac_f.cpp -o $OUT/obj/ransac_f_trace.o
objs=""; for o in *.o; do [ $o = ransac_f.o ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmodsx_trace.so $objs $OUT/obj/ransac_f_trace.o -ldl
echo "built $OUT/oracle/_ref/libdegensac_ref.so (tracing reference) and $OUT/libmodsx_trace.so"

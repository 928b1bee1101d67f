#!/bin/bash
# timing-only knock-out variants of k_match_pack (tools/build_variant.sh pk<N> "-DPACK_KO=<N>"); prints the kernel's average per variant
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/rp_$v
  MODSX_LIB=$R/mods_amd/libmodsx_$v.so rocprofv3 --kernel-trace --stats -d /tmp/rp_$v -o p -- python $R/tools/bench_match.py --tilts 1,2,4,6,8 --phi 120 --reps 10 > /tmp/rp_$v.log 2>&1
  DB=$(find /tmp/rp_$v -name "*.db" | head -1)
  echo "== $v"; python $R/tools/rocprof_summary.py $DB /tmp/s_$v.txt "x" | grep -i "k_match"
done

#!/bin/bash
# build_variant.sh <name> "<extra hipcc flags>"  ->  mods_amd/libmodsx_<name>.so  (kernel experiments; use with MODSX_LIB=...)
# KM_SRC=<file> compiles that file in place of kernels_match.hip (an experimental copy outside the tree)
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../mods_amd/csrc"
out=/tmp/variant_$name; mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -w -I../../include $flags"
for f in kernels_pyramid kernels_affine kernels_orient kernels_describe kernels_views kernels_cand engine engine_views engine_shard capi; do /opt/rocm/bin/hipcc $F -c $f.hip -o $out/$f.o & done
/opt/rocm/bin/hipcc $F -I. -mllvm -amdgpu-mfma-vgpr-form=1 -c ${KM_SRC:-kernels_match.hip} -o $out/kernels_match.o &
for f in ransac ransac_shim ransac_f filters keyfile mser tables; do /opt/rocm/bin/hipcc $F -x hip -c $f.cpp -o $out/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmodsx_$name.so $out/*.o -ldl
ls -la ../libmodsx_$name.so

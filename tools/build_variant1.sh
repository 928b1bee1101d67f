#!/bin/bash
# build_variant1.sh <name> <file.hip> "<extra hipcc flags>"  ->  mods_amd/libmodsx_<name>.so: the tree's objects with ONE kernel
# file recompiled under extra flags (kernel experiments; use with MODSX_LIB=...).  Run `make` first.
# SRC=<path> compiles that file in place of <file.hip> (an experimental copy outside the tree).
set -e
name=$1; file=$2; flags=$3
cd "$(dirname "$0")/../mods_amd/csrc"
out=/tmp/variant1_$name; mkdir -p $out
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -w -I../../include -I. $flags"
base=${file%.hip}
EXTRA=""; [ "$base" = kernels_match ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc $F $EXTRA -c ${SRC:-$file} -o $out/$base.o
objs=""
for o in *.o; do if [ "$o" = "$base.o" ]; then objs="$objs $out/$base.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmodsx_$name.so $objs -ldl
ls -la ../libmodsx_$name.so

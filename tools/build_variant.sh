#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> [-DFLAG ...]  -- libshg with one source rebuilt under extra flags: grates_amd/lib/exp/libshg_<name>.so
# (A/B timing of kernel variants on one box; never loaded by the package itself)
set -e
name=$1; src=$2; shift; shift
cd "$(dirname "$0")/../grates_amd/csrc"
mkdir -p build ../lib/exp
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $( [ "$base" = synthesis_rot ] && echo "-mllvm -amdgpu-mfma-vgpr-form=1" ) "$@" -c $src -o build/${base}_$name.o
objs=$(ls build/*.o | grep -v "_timeline.o" | grep -v "build/${base}\.o" | grep -v "build/${base}_" | grep -v "_[a-zA-Z0-9]*\.o$" || true)
# plain objects of the library (one per source), the variant in place of its source's object
plain=""
for f in plan synthesis synthesis_fused synthesis_rot synthesis_fused32 tables gemm covprop covsep filters points analysis blas gemm_tall blockchol timeseries; do
  if [ "$f" = "$base" ]; then plain="$plain build/${base}_$name.o"; else plain="$plain build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/exp/libshg_$name.so $plain
echo built grates_amd/lib/exp/libshg_$name.so

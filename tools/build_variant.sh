#!/bin/bash
# kernel experiments: build_variants/lib_<name>.so from the working tree with extra hipcc flags, then
#   BN254_LIB_PATH=build_variants/lib_<name>.so python bench.py --no-cpu-baseline --no-host-api
# Only the units named in UNITS are recompiled with the flags (default: the lane-pair kernel file, where most experiments are);
# the other objects come from the last regular build (bn_amd/csrc/build/).
# usage: [UNITS="bn254_kernels_b bn254_kernels_mul"] tools/build_variant.sh NAME [extra hipcc flags...]
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
name=$1; shift
B=bn_amd/csrc/build
UNITS=${UNITS:-bn254_kernels_b}
objs=""
for u in bn254_hip bn254_kernels_b bn254_kernels_mul bn254_kernels_w bn254_kernels_q bn254_multi bn254_measure; do
  if [[ " $UNITS " == *" $u "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-dpp-combine=false "$@" -c bn_amd/csrc/$u.hip -o build_variants/${u}_$name.o &
    objs="$objs build_variants/${u}_$name.o"
  else
    objs="$objs $B/$u.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -lpthread -o build_variants/lib_$name.so
rm -f build_variants/*_$name.o
echo built build_variants/lib_$name.so

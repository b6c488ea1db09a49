#!/bin/bash
# kernel experiments: build_variants/lib_<name>.so from the working tree with extra hipcc flags, then
#   BN254_LIB_PATH=build_variants/lib_<name>.so python bench.py --no-cpu-baseline
# usage: tools/build_variant.sh NAME [extra hipcc flags...]
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value "$@" \
    bn_amd/csrc/bn254_hip.hip bn_amd/csrc/bn254_kernels_b.hip bn_amd/csrc/bn254_kernels_mul.hip bn_amd/csrc/bn254_multi.hip -ldl -lpthread -o build_variants/lib_$name.so
echo built build_variants/lib_$name.so

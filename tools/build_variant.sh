#!/bin/bash
# kernel experiments: build_variants/lib_<name>.so from the working tree with extra hipcc flags, then
#   BN254_LIB_PATH=build_variants/lib_<name>.so python bench.py --no-cpu-baseline --no-host-api
# Only the lane-pair kernel file is recompiled (that is where the experiments are); the other objects come from the last
# regular build (bn_amd/csrc/build/).  usage: tools/build_variant.sh NAME [extra hipcc flags...]
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
name=$1; shift
B=bn_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c bn_amd/csrc/bn254_kernels_b.hip -o build_variants/kb_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/bn254_hip.o build_variants/kb_$name.o $B/bn254_kernels_mul.o $B/bn254_kernels_w.o $B/bn254_multi.o -ldl -lpthread -o build_variants/lib_$name.so
rm -f build_variants/kb_$name.o
echo built build_variants/lib_$name.so

#!/bin/bash
# Builds build_variants/lib_<name>.so from an EDITED device assembly of ONE kernel unit (the rest: shipped objects): assemble -> lld -> bundle ->
# host compile with the fat binary embedded -> link.  usage: tools/variants/asm_variant.sh NAME UNIT file.s   (UNIT e.g. bn254_kernels_q)
# (How a .s is obtained: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value [flags] --cuda-device-only -S bn_amd/csrc/UNIT.hip -o file.s)
set -e
cd "$(dirname "$0")/../.."
name=$1; unit=$2; asm=$3
L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
$L/clang -x assembler --target=amdgcn-amd-amdhsa -mcpu=gfx950 -c "$asm" -o $T/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c bn_amd/csrc/$unit.hip -o $T/$unit.o
objs=""
for u in bn254_hip bn254_kernels_b bn254_kernels_mul bn254_kernels_w bn254_kernels_q bn254_multi bn254_measure; do
  if [ $u == $unit ]; then objs="$objs $T/$unit.o"; else objs="$objs bn_amd/csrc/build/$u.o"; fi
done
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -lpthread -o build_variants/lib_$name.so
rm -rf $T
echo built build_variants/lib_$name.so

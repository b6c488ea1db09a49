#!/bin/bash
# Runs ON THE GPU BOX (round 5, first session): the occupancy question measured (real-mix stream at 1-4 waves per SIMD; a real kernel at
# 2 and 3 waves), the cost of -mllvm -amdgpu-dpp-combine=false on the headline, a baseline bench line.
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibn_amd/csrc tools/ubench_mix.hip -o /tmp/ubench_mix 2> $out/r05a_ubench_build.log
timeout 300 /tmp/ubench_mix 2000 > $out/r05_ubench_mix_occupancy.txt 2>&1; cat $out/r05_ubench_mix_occupancy.txt
for so in bn_amd/libbn254_hip.so build_variants/occ/lib_w3.so build_variants/occ/lib_w3fair0.so; do
  BN254_LIB_PATH=$repo/$so timeout 300 python tools/occupancy_ab.py >> $out/r05_occupancy_ab.jsonl 2>> $out/r05a_err.log
done
cat $out/r05_occupancy_ab.jsonl
tools/ab_bench.sh r05a_dppcombine 3

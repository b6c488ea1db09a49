#!/bin/bash
# Runs ON THE GPU BOX: is the chip power-limited under the pairing kernels?  Samples rocm-smi (average socket power, sclk) while bench.py loops
# 2^16-pairing steps, and while the pure multiply-add stream of tools/ubench_mix (the squaring stream at 1..4 waves per SIMD) runs.
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
log=$out/r05_power_probe.txt; : > $log
sample() { for i in $(seq $2); do echo "== $1 sample $i" >> $log; rocm-smi --showpower --showclocks --showmaxpower --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (edge|junction)" >> $log; sleep 0.3; done; }
echo "---- idle" >> $log; sample idle 2
echo "---- bench.py: 2^16 pairings per step, 1500 steps (Miller + final exponentiation back to back)" >> $log
python bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-host-api --no-side > $out/r05_power_bench.json 2>/dev/null &
pid=$!; sleep 6; sample pairing 8; wait $pid
python tools/brief_line.py < $out/r05_power_bench.json >> $log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibn_amd/csrc tools/ubench_mix.hip -o /tmp/ubench_mix 2>/dev/null
echo "---- tools/ubench_mix 60000 iterations (squaring stream at 1, 2, 3, 4 waves per SIMD, then the dense product)" >> $log
/tmp/ubench_mix 60000 > $out/r05_power_ubench.txt 2>&1 &
pid=$!; sleep 2; sample ubench_mix 24; wait $pid
cat $out/r05_power_ubench.txt >> $log
cat $log

#!/bin/bash
# Runs ON THE GPU BOX: alternates bench.py --batch N over the libraries in build_variants/.  usage: tools/ab_mid.sh TAG "sizes" [rounds]
tag=${1:-abmid}; sizes=${2:-"8192 16384"}; rounds=${3:-3}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
for r in $(seq $rounds); do for so in build_variants/lib_*.so; do for n in $sizes; do
  echo -n "$(basename $so) $n " >> $out/${tag}_ab.txt
  BN254_LIB_PATH=$repo/$so timeout 300 python bench.py --batch $n --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-side 2>/dev/null | python tools/brief_line.py >> $out/${tag}_ab.txt
done; done; done
sort $out/${tag}_ab.txt

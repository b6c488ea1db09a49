#!/bin/bash
# Runs ON THE GPU BOX: alternates bench.py over the libraries in build_variants/ (A B A B ...) so that box-to-box and
# run-to-run drift cancels.  usage: tools/ab_bench.sh TAG [rounds]
tag=${1:-ab}; rounds=${2:-3}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
for r in $(seq $rounds); do
  for so in build_variants/lib_*.so; do
    echo -n "$(basename $so) " >> $out/${tag}_ab.txt
    BN254_LIB_PATH=$repo/$so timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-api --no-side 2>&1 | python tools/brief_line.py >> $out/${tag}_ab.txt
  done
done
sort $out/${tag}_ab.txt

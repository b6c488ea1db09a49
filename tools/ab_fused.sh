#!/bin/bash
repo=$(pwd); out=$repo/gpurun_out
for r in 1 2 3; do
  for v in "a_head 0" "b_new 0" "b_new 1"; do set -- $v
    echo -n "$1 fused=$2 " >> $out/r03a_ab.txt
    BN254_FUSED=$2 BN254_LIB_PATH=$repo/build_variants/lib_$1.so timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-api 2>&1 | grep -E '^\{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.4g ms/step %.3f' % (d['value'], d['ms_per_step']))" >> $out/r03a_ab.txt 2>&1
  done
done
sort $out/r03a_ab.txt

#!/bin/bash
# Runs ON THE GPU BOX (round 4, session e): shared-accumulator Miller loop with the P_i in LDS against the round-3 layout (all state in
# global memory), its tests, and the quad final exponentiation after the scratch fix.
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
tag=r04e
for s in ${@:-tests ab mid}; do
  case $s in
    tests) timeout 1200 python -m pytest tests -m gpu -x -q -k "shared or product or config4 or quad or golden" > $out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_summary.txt; tail -4 $out/${tag}_tests.log | tee -a $out/${tag}_summary.txt ;;
    ab) for r in 1 2 3; do for lib in main pglobal; do
          echo -n "$lib product " >> $out/${tag}_ab.txt
          BN254_LIB_PATH=$repo/build_variants/lib_$lib.so timeout 300 python bench.py --workload product --steps 10 --warmup 2 2>> $out/${tag}_ab.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.5g ms/step %.4f kernels %s' % (d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})); break
else: print('no line')" >> $out/${tag}_ab.txt
        done; done; sort $out/${tag}_ab.txt | tee -a $out/${tag}_summary.txt ;;
    mid) for n in 8192 16384; do timeout 300 python bench.py --batch $n --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-side 2>> $out/${tag}_mid.err | python tools/brief_line.py | tee -a $out/${tag}_summary.txt; done ;;
  esac
done

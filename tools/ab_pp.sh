#!/bin/bash
# Runs ON THE GPU BOX: alternates `bench.py --workload product_prepared` over the libraries in build_variants/ at 2^18 (four pairs per accumulator) and
# 2^17 pairs (two).  usage: tools/ab_pp.sh TAG [rounds] [extra bench flags]
tag=${1:-abpp}; rounds=${2:-2}; shift; shift
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
for r in $(seq $rounds); do for so in build_variants/lib_*.so; do for b in 262144 131072; do
  echo -n "$(basename $so) $b " >> $out/${tag}_ab.txt
  BN254_LIB_PATH=$repo/$so timeout 300 python bench.py --workload product_prepared --batch $b --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        print('value %.5g ms/step %.4f kernels %s' % (d['value'], d['ms_per_step'], {k: (round(v,4) if not isinstance(v,dict) else {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("power_W","sclk_MHz","frac","frac_executed","table_GBps")}) for k,v in d["kernel_ms_per_step"].items()})); break
else: print('no line')" >> $out/${tag}_ab.txt
done; done; done
sort $out/${tag}_ab.txt

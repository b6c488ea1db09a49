#!/bin/bash
# Runs ON THE GPU BOX: alternates bench.py side workloads over the libraries in build_variants/.  usage: tools/ab_side.sh TAG "workloads" [rounds]
tag=${1:-abside}; wl=${2:-"product gtpow prepared"}; rounds=${3:-2}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
for r in $(seq $rounds); do for so in build_variants/lib_*.so; do for w in $wl; do
  echo -n "$(basename $so) $w " >> $out/${tag}_ab.txt
  BN254_LIB_PATH=$repo/$so timeout 300 python bench.py --workload $w --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('value %.5g ms/step %.4f kernels %s' % (d['value'], d['ms_per_step'], {k: round(v['avg_launch_ms'],4) for k,v in r.get('kernels',{}).items()} or {k: round(v,4) for k,v in (d.get('kernel_ms_per_step') or d.get('kernel_ms') or {}).items()})); break
else: print('no line')" >> $out/${tag}_ab.txt
done; done; done
sort $out/${tag}_ab.txt

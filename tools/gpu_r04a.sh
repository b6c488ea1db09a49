#!/bin/bash
# Runs ON THE GPU BOX (round 4, session a): the DFMA leaf experiment, the baseline line, the traffic-cost A/B (lib_main against
# lib_alias: every table / state buffer aliased onto a cache-resident footprint - same instruction stream, no HBM traffic, wrong
# results) and the PMC passes of the G1 kernel after the cold-call fix.   usage: tools/gpu_r04a.sh [steps...]
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
tag=r04a
steps=${@:-dfma tests bench ab pmc}
for s in $steps; do
  case $s in
    dfma) timeout 300 build_variants/dfma_experiment > $out/${tag}_dfma.txt 2>&1; echo "dfma rc=$?" | tee -a $out/${tag}_summary.txt; cat $out/${tag}_dfma.txt | tee -a $out/${tag}_summary.txt ;;
    tests) timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scalar_mul or group_addition or config5 or golden_fixtures or full_size_bilinearity or gt_pow_modes" > $out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_summary.txt; tail -3 $out/${tag}_tests.log | tee -a $out/${tag}_summary.txt ;;
    bench) timeout 600 python bench.py --steps 20 --warmup 3 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?" | tee -a $out/${tag}_summary.txt; python tools/brief_line.py < $out/${tag}_bench.json | tee -a $out/${tag}_summary.txt ;;
    ab) for r in 1 2; do for lib in main alias; do
          for w in pairing g1mul g2mul gtpow product; do
            extra="--no-cpu-baseline --no-host-api --no-side"
            echo -n "$lib $w " >> $out/${tag}_ab.txt
            BN254_LIB_PATH=$repo/build_variants/lib_$lib.so timeout 300 python bench.py --workload $w --steps 10 --warmup 2 $extra 2>> $out/${tag}_ab.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print('value %.5g ms/step %.4f kernels %s' % (d['value'], d['ms_per_step'], {k: round(v['avg_launch_ms'],4) for k,v in r.get('kernels',{}).items()} or d.get('kernel_ms_per_step')))
        break
else: print('no line')
" >> $out/${tag}_ab.txt
          done; done; done
        sort $out/${tag}_ab.txt | tee -a $out/${tag}_summary.txt ;;
    pmc) cd /tmp; for w in g1mul g2mul; do for c in FETCH_SIZE WRITE_SIZE; do
            timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_${w}_$c -- python $repo/bench.py --workload $w --steps 2 --warmup 1 > $out/${tag}_pmc_${w}_$c.log 2>&1; done; done
          find $out -name "*.db" -delete 2>/dev/null; cd $repo
          python tools/summarize_pmc_side.py $tag 2>&1 | tee -a $out/${tag}_summary.txt ;;
  esac
done
du -sh $out | tail -1

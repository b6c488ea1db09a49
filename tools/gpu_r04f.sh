#!/bin/bash
# Runs ON THE GPU BOX (round 4, session f): role-merged reductions in the four-lane squaring / product - full suite, latency table, mid-size lines.
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
tag=r04f
for s in ${@:-tests mid latency}; do
  case $s in
    tests) timeout 2400 python -m pytest tests -m gpu -x -q > $out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_summary.txt; tail -4 $out/${tag}_tests.log | tee -a $out/${tag}_summary.txt ;;
    mid) for n in 8192 16384; do timeout 300 python bench.py --batch $n --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-side 2>> $out/${tag}_mid.err > $out/${tag}_bench_$n.json; python tools/brief_line.py < $out/${tag}_bench_$n.json | tee -a $out/${tag}_summary.txt; done ;;
    latency) timeout 900 python tools/wave_latency.py > $out/${tag}_latency.json 2> $out/${tag}_latency.err; echo "latency rc=$?" | tee -a $out/${tag}_summary.txt; python -c "
import json; d=json.load(open('$out/${tag}_latency.json')); print(json.dumps(d['pairing_batch_ms_mid_size']['default'])); print(json.dumps(d['pairing_batch_ms_mid_size']['quad'])); print(json.dumps(d['final_exp_ms']))" | tee -a $out/${tag}_summary.txt ;;
  esac
done

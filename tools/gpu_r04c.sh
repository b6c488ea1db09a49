#!/bin/bash
# Runs ON THE GPU BOX (round 4, session c): full GPU suite, the G1 scalar-multiplication A/B (dual-product formulas in all; window table
# packed / one 128-byte line per entry / prefetched a window ahead), bench line, latency table with the retuned thresholds.
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
tag=r04c
steps=${@:-tests ab bench latency}
for s in $steps; do
  case $s in
    tests) timeout 2400 python -m pytest tests -m gpu -x -q > $out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_summary.txt; tail -12 $out/${tag}_tests.log | tee -a $out/${tag}_summary.txt ;;
    ab) for r in 1 2 3; do for lib in main t128 t128pf pf; do
          echo -n "$lib g1mul " >> $out/${tag}_ab.txt
          BN254_LIB_PATH=$repo/build_variants/lib_$lib.so timeout 300 python bench.py --workload g1mul --steps 10 --warmup 2 2>> $out/${tag}_ab.err | python tools/brief_line.py >> $out/${tag}_ab.txt
        done; done; sort $out/${tag}_ab.txt | tee -a $out/${tag}_summary.txt ;;
    bench) timeout 900 python bench.py --steps 20 --warmup 3 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?" | tee -a $out/${tag}_summary.txt; python tools/brief_line.py < $out/${tag}_bench.json | tee -a $out/${tag}_summary.txt
           timeout 300 python bench.py --batch 8192 --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-side > $out/${tag}_bench_8192.json 2>> $out/${tag}_bench.err; python tools/brief_line.py < $out/${tag}_bench_8192.json | tee -a $out/${tag}_summary.txt ;;
    latency) timeout 900 python tools/wave_latency.py > $out/${tag}_latency.json 2> $out/${tag}_latency.err; echo "latency rc=$?" | tee -a $out/${tag}_summary.txt; python -c "
import json; d=json.load(open('$out/${tag}_latency.json')); print(json.dumps(d['pairing_batch_ms_mid_size']['default']))" | tee -a $out/${tag}_summary.txt ;;
  esac
done

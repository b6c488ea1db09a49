#!/bin/bash
# Runs ON THE GPU BOX (round 6, first contact of the native prepared-G2 path): its tests, the prepared bench line in both modes.
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
tag=r06a
timeout 1200 python -m pytest tests/test_gpu_prepared_native.py -x -q > $out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_summary.txt; tail -15 $out/${tag}_tests.log | tee -a $out/${tag}_summary.txt
for m in native reference native reference; do timeout 300 python bench.py --workload prepared --prepared-mode $m --steps 20 --warmup 3 >> $out/${tag}_prepared.json 2>> $out/${tag}_bench.err; done
python - <<PY | tee -a $out/${tag}_summary.txt
import json
for l in open('$out/${tag}_prepared.json'):
    d=json.loads(l); print(d['config']['prepared_mode'], '%.4g' % d['value'], d['unit'], '%.3f ms/step' % d['ms_per_step'], d['kernel_ms'], 'frac %.3f' % d['roofline']['frac'])
PY
tail -5 $out/${tag}_bench.err

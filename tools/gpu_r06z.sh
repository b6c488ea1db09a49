#!/bin/bash
# Runs ON THE GPU BOX (round 6): smoke(), the full GPU suite, the bench line (with its power leg), rocprofv3 --kernel-trace --stats of the SAME
# command, the PMC passes of every kernel incl. the native prepared Miller loop (profiles/pmc_traffic.json), the latency table.
# usage: tools/gpu_r06z.sh TAG [steps...]
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
tag=${1:-r06z}; shift
for s in ${@:-smoke tests bench prof sq latency}; do
  case $s in
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/${tag}_summary.txt; tail -2 $out/${tag}_smoke.log | tee -a $out/${tag}_summary.txt ;;
    tests) timeout 3000 python -m pytest tests -m gpu -x -q > $out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_summary.txt; tail -4 $out/${tag}_tests.log | tee -a $out/${tag}_summary.txt ;;
    newtests) timeout 1200 python -m pytest tests/test_gpu_prepared_native.py -x -q > $out/${tag}_newtests.log 2>&1; echo "newtests rc=$?" | tee -a $out/${tag}_summary.txt; tail -4 $out/${tag}_newtests.log | tee -a $out/${tag}_summary.txt ;;
    bench) timeout 900 python bench.py --steps 20 --warmup 3 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?" | tee -a $out/${tag}_summary.txt; python tools/brief_line.py < $out/${tag}_bench.json | tee -a $out/${tag}_summary.txt
           timeout 900 python bench.py > $out/${tag}_bench_default_flags.json 2>> $out/${tag}_bench.err; python tools/brief_line.py < $out/${tag}_bench_default_flags.json | tee -a $out/${tag}_summary.txt
           for w in g1mul g2mul gtpow product prepared product_prepared; do timeout 300 python bench.py --workload $w --steps 10 --warmup 2 >> $out/${tag}_side.json 2>> $out/${tag}_bench.err; done
           timeout 300 python bench.py --workload prepared --prepared-mode reference --steps 10 --warmup 2 >> $out/${tag}_side.json 2>> $out/${tag}_bench.err
           timeout 300 python bench.py --workload prepared --prepared-mode native_per_q --steps 10 --warmup 2 >> $out/${tag}_side.json 2>> $out/${tag}_bench.err
           python -c "
import json
for l in open('$out/${tag}_side.json'):
    d=json.loads(l); print(d['metric'][:40], d['config'].get('prepared_mode',''), '%.4g' % d['value'], d['unit'], '%.3f ms/step' % d['ms_per_step'])" | tee -a $out/${tag}_summary.txt ;;
    prof) cd /tmp
          timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python $repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-side > $out/${tag}_stats.log 2>&1
          timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_statsside -- python $repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-api > $out/${tag}_statsside.log 2>&1
          timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_statsprep -- python $repo/bench.py --workload prepared --steps 20 --warmup 3 > $out/${tag}_statsprep.log 2>&1
          find $out/${tag}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_kernel_stats.csv
          find $out/${tag}_statsside -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_with_side_kernel_stats.csv
          find $out/${tag}_statsprep -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_prepared_kernel_stats.csv
          grep -h "^{" $out/${tag}_stats.log > $out/${tag}_bench_line_under_rocprof.json
          grep -h "^{" $out/${tag}_statsprep.log > $out/${tag}_prepared_line_under_rocprof.json
          for c in FETCH_SIZE WRITE_SIZE; do
            timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_pairing_$c -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-side --no-power > $out/${tag}_pmc_pairing_$c.log 2>&1
            for w in g1mul g2mul gtpow product prepared product_prepared; do
              timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_${w}_$c -- python $repo/bench.py --workload $w --steps 2 --warmup 1 > $out/${tag}_pmc_${w}_$c.log 2>&1
            done
            timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_prepq_$c -- python $repo/bench.py --workload prepared --prepared-mode native_per_q --steps 2 --warmup 1 > $out/${tag}_pmc_prepq_$c.log 2>&1
          done
          find $out -name "*.db" -delete 2>/dev/null; cd $repo
          python tools/summarize_pmc_all.py $tag $out/${tag}_pmc_* --gt_product=65536 2>&1 | tee -a $out/${tag}_summary.txt
          cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json; cp profiles/${tag}_pmc_all.txt $out/ ;;
    pmcprep) cd /tmp
          for c in FETCH_SIZE WRITE_SIZE; do
            timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_prepared_$c -- python $repo/bench.py --workload prepared --steps 2 --warmup 1 > $out/${tag}_pmc_prepared_$c.log 2>&1
            timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_prepq_$c -- python $repo/bench.py --workload prepared --prepared-mode native_per_q --steps 2 --warmup 1 > $out/${tag}_pmc_prepq_$c.log 2>&1
          done
          find $out -name "*.db" -delete 2>/dev/null; cd $repo
          python tools/summarize_pmc_all.py $tag $out/${tag}_pmc_* 2>&1 | tee -a $out/${tag}_summary.txt
          cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json; cp profiles/${tag}_pmc_all.txt $out/ ;;
    sq) kms=$(python -c "
import json
d=json.loads(open('$out/${tag}_bench.json').readline()); k=d['roofline']['kernels']
print('miller=%.4f,final_exp=%.4f' % (k['miller']['avg_launch_ms'], k['final_exp']['avg_launch_ms']))")
        cd /tmp
        for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS; do
          timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_sq_$c -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-side --no-power > $out/${tag}_sq_$c.log 2>&1
        done
        find $out -name "*.db" -delete 2>/dev/null; cd $repo
        python tools/summarize_sq.py $tag $out/${tag}_sq_* --kernel-ms=$kms | tee $out/${tag}_sq_counters.txt; cp profiles/sq_counters.json $out/${tag}_sq_counters.json ;;
    latency) timeout 900 python tools/wave_latency.py > $out/${tag}_latency.json 2> $out/${tag}_latency.err; echo "latency rc=$?" | tee -a $out/${tag}_summary.txt ;;
  esac
done

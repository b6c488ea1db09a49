#!/bin/bash
# Runs ON THE GPU BOX (round 4, session b): the whole GPU test suite with the round's new pieces (options API, device KATs, quad
# kernels, multi_c bench), the latency table of the mid-size window, the bench line, the Gt::pow traffic A/B and the PMC / kernel-trace
# passes behind profiles/pmc_traffic.json.   usage: tools/gpu_r04b.sh [steps...]
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
tag=r04b
steps=${@:-tests latency bench ab prof}
for s in $steps; do
  case $s in
    newtests) timeout 1200 python -m pytest tests -m gpu -x -q -k "quad or kat or options or multi_c or golden or wave_final or wave_pairing or shared_acc or gt_pow_modes or reference" > $out/${tag}_newtests.log 2>&1; echo "newtests rc=$?" | tee -a $out/${tag}_summary.txt; tail -15 $out/${tag}_newtests.log | tee -a $out/${tag}_summary.txt ;;
    tests) timeout 2400 python -m pytest tests -m gpu -x -q > $out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_summary.txt; tail -15 $out/${tag}_tests.log | tee -a $out/${tag}_summary.txt ;;
    latency) timeout 900 python tools/wave_latency.py > $out/${tag}_latency.json 2> $out/${tag}_latency.err; echo "latency rc=$?" | tee -a $out/${tag}_summary.txt; python -c "
import json; d=json.load(open('$out/${tag}_latency.json')); print(json.dumps(d['pairing_batch_ms_mid_size'], indent=0))" | tee -a $out/${tag}_summary.txt; tail -3 $out/${tag}_latency.err ;;
    bench) timeout 900 python bench.py --steps 20 --warmup 3 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?" | tee -a $out/${tag}_summary.txt; python tools/brief_line.py < $out/${tag}_bench.json | tee -a $out/${tag}_summary.txt
           timeout 300 python bench.py --batch 8192 --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-side > $out/${tag}_bench_8192.json 2>> $out/${tag}_bench.err; python tools/brief_line.py < $out/${tag}_bench_8192.json | tee -a $out/${tag}_summary.txt
           BN254_BENCH_SHARE_GPU=1 timeout 600 python bench.py --mode multi_c --gpus 2 --steps 3 --warmup 1 > $out/${tag}_bench_multi_c.json 2>> $out/${tag}_bench.err; head -c 1500 $out/${tag}_bench_multi_c.json | tee -a $out/${tag}_summary.txt ;;
    ab) for r in 1 2; do for lib in main alias; do
          echo -n "$lib gtpow " >> $out/${tag}_ab.txt
          BN254_LIB_PATH=$repo/build_variants/lib_$lib.so timeout 300 python bench.py --workload gtpow --steps 10 --warmup 2 2>> $out/${tag}_ab.err | python tools/brief_line.py >> $out/${tag}_ab.txt
        done; done; sort $out/${tag}_ab.txt | tee -a $out/${tag}_summary.txt ;;
    prof) cd /tmp
          timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python $repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-side > $out/${tag}_stats.log 2>&1
          timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_statsside -- python $repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-api > $out/${tag}_statsside.log 2>&1
          find $out/${tag}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_kernel_stats.csv
          find $out/${tag}_statsside -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_with_side_kernel_stats.csv
          for c in FETCH_SIZE WRITE_SIZE; do
            timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_pairing_$c -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-side > $out/${tag}_pmc_pairing_$c.log 2>&1
            timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_mid_$c -- python $repo/bench.py --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-side > $out/${tag}_pmc_mid_$c.log 2>&1
            timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_small_$c -- python $repo/bench.py --batch 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-side > $out/${tag}_pmc_small_$c.log 2>&1
            for w in g1mul g2mul gtpow product; do
              timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_${w}_$c -- python $repo/bench.py --workload $w --steps 2 --warmup 1 > $out/${tag}_pmc_${w}_$c.log 2>&1
            done
          done
          find $out -name "*.db" -delete 2>/dev/null; cd $repo
          python tools/summarize_pmc_all.py $tag $out/${tag}_pmc_* --gt_product=65536 2>&1 | tee -a $out/${tag}_summary.txt
          cp profiles/pmc_traffic.json $out/${tag}_pmc_traffic.json; cp profiles/${tag}_pmc_all.txt $out/ ;;
  esac
done
du -sh $out | tail -1

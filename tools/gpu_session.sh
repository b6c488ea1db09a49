#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the whole GPU suite of a round in one call, every step under its own timeout,
# everything logged under gpurun_out/<tag>_*.  usage: tools/gpu_session.sh TAG [steps...]   (default steps: tests bench side)
tag=${1:-sess}; shift
steps=${@:-tests bench side}
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out
export TMPDIR=/tmp
for s in $steps; do
  case $s in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q > $out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a $out/${tag}_summary.txt; tail -3 $out/${tag}_tests.log | tee -a $out/${tag}_summary.txt ;;
    wavetests) timeout 900 python -m pytest tests/test_gpu_wave.py -x -q > $out/${tag}_wavetests.log 2>&1; echo "wavetests rc=$?" | tee -a $out/${tag}_summary.txt; tail -15 $out/${tag}_wavetests.log | tee -a $out/${tag}_summary.txt ;;
    latency) timeout 600 python tools/wave_latency.py > $out/${tag}_latency.json 2> $out/${tag}_latency.err; echo "latency rc=$?" | tee -a $out/${tag}_summary.txt; cat $out/${tag}_latency.json | tee -a $out/${tag}_summary.txt; tail -5 $out/${tag}_latency.err ;;
    bench) timeout 600 python bench.py --steps 20 --warmup 3 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?" | tee -a $out/${tag}_summary.txt; cat $out/${tag}_bench.json | tee -a $out/${tag}_summary.txt ;;
    side) for w in g1mul g2mul gtpow product prepared; do timeout 300 python bench.py --workload $w --steps 5 --warmup 1 >> $out/${tag}_side.json 2>> $out/${tag}_side.err; done; cat $out/${tag}_side.json | tee -a $out/${tag}_summary.txt ;;
    hostapi) for cfg in "2 0" "1 0" "2 32768" "4 16384"; do set -- $cfg; echo "slots=$1 chunk=$2" >> $out/${tag}_hostapi.txt; BN254_PIPELINE_SLOTS=$1 BN254_PIPELINE_CHUNK=$2 timeout 300 python tools/host_api_rate.py >> $out/${tag}_hostapi.txt 2>&1; done; timeout 300 python tools/host_api_rate.py 1048576 >> $out/${tag}_hostapi.txt 2>&1; cat $out/${tag}_hostapi.txt | tee -a $out/${tag}_summary.txt ;;
    prof) cd /tmp; B="python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-side"
          timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python $repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-api --no-side > $out/${tag}_stats.log 2>&1
          timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_statsside -- python $repo/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-api > $out/${tag}_statsside.log 2>&1
          find $out/${tag}_statsside -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_with_side_kernel_stats.csv
          timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- $B > $out/${tag}_fetch.log 2>&1
          timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- $B > $out/${tag}_write.log 2>&1
          timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/${tag}_sq -- $B > $out/${tag}_sq.log 2>&1
          find $out/${tag}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_kernel_stats.csv
          find $out -name "*.db" -delete 2>/dev/null; cd $repo ;;
    pmcside) cd /tmp; for w in g1mul g2mul gtpow product; do for c in FETCH_SIZE WRITE_SIZE; do
            timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_${w}_$c -- python $repo/bench.py --workload $w --steps 2 --warmup 1 > $out/${tag}_pmc_${w}_$c.log 2>&1; done; done
          find $out -name "*.db" -delete 2>/dev/null; cd $repo ;;
    variants) for so in build_variants/lib_*.so; do echo "== $so" >> $out/${tag}_variants.txt; BN254_LIB_PATH=$repo/$so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-api 2>&1 | python tools/brief_line.py >> $out/${tag}_variants.txt; done; cat $out/${tag}_variants.txt | tee -a $out/${tag}_summary.txt ;;
  esac
done
du -sh $out | tail -1

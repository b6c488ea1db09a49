#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the headline kernels, one rocprofv3 --pmc pass per counter (kernel trace off), 3 timed steps of 2^16 pairings.
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_SALU; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/r04s_pmc_$c -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-side > $out/r04s_pmc_$c.log 2>&1
done
find $out -name "*.db" -delete 2>/dev/null
cd $repo; python tools/summarize_sq.py r04s $out/r04s_pmc_* | tee $out/r04s_sq.txt

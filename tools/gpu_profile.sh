#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + the separate PMC passes the roofline fields are built from.
# usage: tools/gpu_profile.sh TAG     -> gpurun_out/<TAG>_{stats,fetch,write,sq}/ ; summarise with tools/summarize_pmc.py
set -u
tag=${1:-prof}
repo=$(pwd)
out=$repo/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- $B > $out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- $B > $out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- $B > $out/${tag}_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/${tag}_sq -- $B > $out/${tag}_sq.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $out/${tag}_ic -- $B > $out/${tag}_ic.log 2>&1
find $out/${tag}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_kernel_stats.csv
# keep only the small csv files (the merge back is capped)
find $out -name "*.db" -delete 2>/dev/null
du -sh $out | tail -1

#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the side workloads (one csv per workload under gpurun_out/<tag>_<w>_kernel_stats.csv)
tag=${1:-side}; repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for w in g1mul g2mul gtpow product prepared; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_$w -- python $repo/bench.py --workload $w --steps 3 --warmup 1 > $out/${tag}_$w.log 2>&1
  find $out/${tag}_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_${w}_kernel_stats.csv
  rm -rf $out/${tag}_$w
done
ls $out | grep $tag

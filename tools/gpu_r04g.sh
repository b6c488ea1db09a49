#!/bin/bash
# round 4, later session: parity of the working tree's library (quick GPU tests), then A/B of build_variants/ on the headline and mid sizes
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; tag=${TAG:-r04g}
for what in "$@"; do case $what in
  tests) timeout 1500 python -m pytest tests -m gpu -x -q -k "${K:-kats or golden or quad or wave or full or reference}" 2>&1 | tail -5 > gpurun_out/${tag}_tests.log; cat gpurun_out/${tag}_tests.log;;
  alltests) timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${tag}_alltests.log; cat gpurun_out/${tag}_alltests.log;;
  ab) rm -f gpurun_out/${tag}_ab.txt; bash tools/ab_bench.sh $tag ${ROUNDS:-3} > /dev/null; sort gpurun_out/${tag}_ab.txt;;
  mid) rm -f gpurun_out/${tag}mid_ab.txt; bash tools/ab_mid.sh ${tag}mid "8192" 2 > /dev/null; sort gpurun_out/${tag}mid_ab.txt;;
  side) rm -f gpurun_out/${tag}side_ab.txt; bash tools/ab_side.sh ${tag}side > /dev/null 2>&1; sort gpurun_out/${tag}side_ab.txt;;
esac; done

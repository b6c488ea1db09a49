#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 --pmc pass per library in build_variants/ over a short bench run.
# usage: tools/pmc_variants.sh TAG "COUNTER COUNTER ..."
tag=$1; counters=$2
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for so in $repo/build_variants/lib_*.so; do
  n=$(basename $so .so)
  BN254_LIB_PATH=$so timeout 300 rocprofv3 --pmc $counters --output-format csv -d $out/${tag}_$n -- python $repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-api --no-side > $out/${tag}_$n.log 2>&1
  python3 - "$out/${tag}_$n" "$n" <<'PY' >> $out/${tag}_pmc_variants.txt
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("::")[-1].split("(")[0]
        if "miller_naf" in k or "final_exp_B" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(sys.argv[2], k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg[k].items())))
PY
done
find $out -name "*.db" -delete 2>/dev/null
cat $out/${tag}_pmc_variants.txt

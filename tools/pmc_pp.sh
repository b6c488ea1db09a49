#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --pmc passes (one counter set each) over `bench.py --workload product_prepared` for the shipped library.
# usage: tools/pmc_pp.sh TAG [extra bench flags]
tag=$1; shift
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/${tag}_p$i -- python $repo/bench.py --workload product_prepared --steps 2 --warmup 1 "$@" > $out/${tag}_p$i.log 2>&1
  python3 - "$out/${tag}_p$i" <<'PY' >> $out/${tag}_pmc.txt
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("::")[-1].split("(")[0]
        if "native_shared" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, " ".join("%s=%.6g(n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(agg[k].items())))
PY
done
find $out -name "*.db" -delete 2>/dev/null
cat $out/${tag}_pmc.txt

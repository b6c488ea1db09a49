#!/bin/bash
# Runs ON THE GPU BOX: SQ stall counters of one kernel over a bench workload.  usage: tools/pmc_kernel.sh TAG WORKLOAD KERNEL_SUBSTRING [extra bench flags]
tag=$1; wl=$2; kn=$3; shift; shift; shift
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/${tag}_p$i -- python $repo/bench.py --workload $wl --steps 2 --warmup 1 "$@" > $out/${tag}_p$i.log 2>&1
  python3 - "$out/${tag}_p$i" "$kn" <<'PY' >> $out/${tag}_pmc.txt
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("::")[-1].split("(")[0]
        if sys.argv[2] in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, " ".join("%s=%.5g" % (c, sum(v) / len(v)) for c, v in sorted(agg[k].items())))
PY
done
find $out -name "*.db" -delete 2>/dev/null
cat $out/${tag}_pmc.txt

#!/bin/bash
# Runs ON THE GPU BOX: for every kernel of a bench workload, the average wave's residency as a share of the kernel's duration
# = (SQ_WAVE_CYCLES x 4 / SQ_WAVES) / (GRBM_GUI_ACTIVE / 8 XCDs).  Near 1: the waves of a SIMD finish together; well below: some waves leave early and
# the rest run alone (how the missing hand-over of Gt::pow was found, round 6).  usage: tools/residency.sh TAG "workload ..." 
tag=$1; wls=$2
repo=$(pwd); out=$repo/gpurun_out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for wl in $wls; do
  extra=""; [ "$wl" = pairing ] && extra="--no-cpu-baseline --no-host-api --no-side --no-power"
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_$wl -- python $repo/bench.py --workload $wl --steps 2 --warmup 1 $extra > $out/${tag}_$wl.log 2>&1
  python3 - "$out/${tag}_$wl" "$wl" <<'PY' >> $out/${tag}_residency.txt
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("::")[-1].split("(")[0]
        agg[k][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    g = max(agg[k]); c = {n: sum(v) / len(v) for n, v in agg[k][g].items()}
    if c.get("SQ_WAVES", 0) < 512 or "GRBM_GUI_ACTIVE" not in c: continue
    res = c["SQ_WAVE_CYCLES"] * 4 / c["SQ_WAVES"]; dur = c["GRBM_GUI_ACTIVE"] / 8
    print("%-14s %-34s waves %6d  wave residency %9.0f cycles  kernel %9.0f cycles  share %.3f" % (sys.argv[2], k, c["SQ_WAVES"], res, dur, res / dur))
PY
done
find $out -name "*.db" -delete 2>/dev/null
cat $out/${tag}_residency.txt

#!/usr/bin/env python3
"""rocprofv3 --pmc passes of SQ counters (one counter per pass) -> per-kernel averages per dispatch and per wave.
usage: tools/summarize_sq.py TAG DIR [DIR ...]"""
import collections, csv, glob, pathlib, sys
tag, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))          # kernel -> counter -> values of the largest grid
grid = {}
for d in dirs:
    for f in glob.glob(str(pathlib.Path(d) / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("::")[-1].split("(")[0]
            if not k.startswith("bn254_") or "ubench" in k: continue
            g = int(r["Grid_Size"]); grid[k] = max(grid.get(k, 0), g)
            agg[k][(r["Counter_Name"], g)].append(float(r["Counter_Value"]))
counters = sorted({c for k in agg for (c, g) in agg[k]})
print(f"session {tag}: rocprofv3 --pmc, one counter per pass; per dispatch of the largest grid (2^16 pairings = 2048 waves of 64 lanes), averaged over the dispatches")
print("%-22s %8s " % ("kernel", "waves") + " ".join("%20s" % c for c in counters))
for k in sorted(agg):
    g = grid[k]; waves = g // 64
    vals = [sum(agg[k][(c, g)]) / max(1, len(agg[k][(c, g)])) if (c, g) in agg[k] else float("nan") for c in counters]
    print("%-22s %8d " % (k, waves) + " ".join("%20.4g" % v for v in vals))
    print("%-22s %8s " % ("  per wave", "") + " ".join("%20.4g" % (v / waves) for v in vals))

#!/usr/bin/env python3
"""Summarise the per-workload rocprofv3 --pmc passes of tools/gpu_session.sh (step pmcside: gpurun_out/<tag>_pmc_<workload>_{FETCH,WRITE}_SIZE)
into profiles/<tag>_pmc_side.txt.   usage: tools/summarize_pmc_side.py TAG"""
import collections, csv, glob, pathlib, sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
tag = sys.argv[1]
SKIP = ("copyBuffer", "synthetic", "tile_k", "ubench", "fillBuffer")

def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(str(d / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("::")[-1].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}

lines = ["rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python bench.py --workload W --steps 2 --warmup 1`; KB per dispatch, averages;",
         "raw = (F + W) KB, corrected = (2F + W) KB (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads 2x on gfx950)",
         "algorithmic bytes per launch: g1mul 2^18 (one sub-launch) x 224 B = 0.059 GB; g2mul 2^17 x 416 B = 0.055 GB; gtpow 2^16 x 800 B = 0.052 GB"]
for w in ("g1mul", "g2mul", "gtpow", "product"):
    F = load(ROOT / "gpurun_out" / f"{tag}_pmc_{w}_FETCH_SIZE", "FETCH_SIZE"); W = load(ROOT / "gpurun_out" / f"{tag}_pmc_{w}_WRITE_SIZE", "WRITE_SIZE")
    for k in sorted(F):
        if any(s in k for s in SKIP) or k not in W: continue
        if "chain" in k or (w == "gtpow" and "gt_pow" not in k) or (w in ("g1mul", "g2mul") and "mul_M" not in k): continue      # input generation
        lines.append("%-10s %-24s FETCH_KB %12.1f WRITE_KB %12.1f  raw_GB %.4f corr_GB %.4f" % (w, k, F[k], W[k], (F[k] + W[k]) * 1024 / 1e9, (2 * F[k] + W[k]) * 1024 / 1e9))
(ROOT / "profiles" / f"{tag}_pmc_side.txt").write_text("\n".join(lines) + "\n")
print("\n".join(lines))

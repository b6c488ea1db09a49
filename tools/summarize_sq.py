#!/usr/bin/env python3
"""rocprofv3 --pmc passes of SQ counters (one counter per pass) -> per-kernel averages per dispatch and per wave.
usage: tools/summarize_sq.py TAG DIR [DIR ...]"""
import collections, csv, glob, json, pathlib, sys
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

# profiles/sq_counters.json: what bench.py quotes beside the multiply-add roofline (kernel name as in bn254_kernel_stats)
NAMES = {"bn254_miller_naf_B": "miller", "bn254_final_exp_B": "final_exp"}
out = {}
for k, name in NAMES.items():
    if k not in agg: continue
    g = grid[k]; waves = g // 64
    def avg(c):
        v = agg[k].get((c, g)); return sum(v) / len(v) if v else None
    valu, salu, gui = avg("SQ_INSTS_VALU"), avg("SQ_INSTS_SALU"), avg("GRBM_GUI_ACTIVE")
    if not (valu and gui): continue
    cyc = gui / 8                                                       # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    simds = 256 * 4
    out[name] = {"kernel": k, "waves": waves, "valu_instructions_per_wave": valu / waves, "salu_instructions_per_wave": (salu or 0) / waves,
                 "busy_cycles_per_xcd": cyc, "cycles_per_valu_instruction_per_simd": cyc / (valu / simds),
                 "issue_limit_cycles": 4, "valu_issue_utilisation": 4 / (cyc / (valu / simds)),
                 "source": f"profiles/{tag}_sq_counters.txt (separate rocprofv3 --pmc passes, NOT this run)"}
if out:
    (pathlib.Path(__file__).resolve().parents[1] / "profiles" / "sq_counters.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")

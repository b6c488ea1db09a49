#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_*/x_counter_collection.csv) into profiles/<tag>_pmc.txt and update
profiles/pmc_traffic.json (read by bench.py for roofline.traffic).
usage: tools/summarize_pmc.py TAG fetch_dir write_dir [sq_dir]"""
import collections, csv, glob, json, pathlib, sys

def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
    for f in glob.glob(str(pathlib.Path(d) / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("::")[-1].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"])
    return agg, meta

def main():
    tag, fd, wd = sys.argv[1:4]
    sq = sys.argv[4] if len(sys.argv) > 4 else None
    root = pathlib.Path(__file__).resolve().parents[1]
    fa, meta = load(fd); wa, _ = load(wd)
    lines = [f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 3 --warmup 1` [{tag}]",
             "FETCH_SIZE/WRITE_SIZE are KB per dispatch.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide coalesced reads 2x,",
             "so HBM bytes per launch are given as a range: raw = (F + W) * 1024, corrected = (2F + W) * 1024.", "",
             "%-22s %14s %14s %12s %12s %6s %6s %8s %8s" % ("kernel", "FETCH_KB", "WRITE_KB", "raw_GB", "corr_GB", "VGPR", "AGPR", "scratch", "grid")]
    traffic = {}
    for k in sorted(fa):
        F = sum(fa[k]["FETCH_SIZE"]) / len(fa[k]["FETCH_SIZE"]); W = sum(wa[k]["WRITE_SIZE"]) / len(wa[k]["WRITE_SIZE"])
        raw = (F + W) * 1024; corr = (2 * F + W) * 1024
        lines.append("%-22s %14.1f %14.1f %12.4f %12.4f %6s %6s %8s %8s" % ((k, F, W, raw / 1e9, corr / 1e9) + meta[k][:4]))
        short = {"bn254_miller_A": "miller", "bn254_miller_B": "miller", "bn254_miller_naf_B": "miller", "bn254_final_exp_A": "final_exp", "bn254_final_exp_B": "final_exp"}.get(k)
        if short:
            traffic[short] = {"hbm_bytes_per_launch": corr, "hbm_bytes_per_launch_raw": raw, "kernel": k, "source": f"profiles/{tag}_pmc.txt"}
    if sq:
        sa, _ = load(sq)
        lines += ["", "SQ counters (one pass), averages per dispatch:"]
        for k in sorted(sa):
            lines.append("  " + k + ": " + ", ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(sa[k].items())))
            c = {n: sum(v) / len(v) for n, v in sa[k].items()}
            if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
                lines.append("    VALU instr per wave = %.0f" % (c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1)))
    (root / "profiles" / f"{tag}_pmc.txt").write_text("\n".join(lines) + "\n")
    (root / "profiles" / "pmc_traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")
    print("\n".join(lines))

if __name__ == "__main__":
    main()

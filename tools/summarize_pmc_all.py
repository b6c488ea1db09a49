#!/usr/bin/env python3
"""All rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a GPU session -> profiles/<tag>_pmc_all.txt and profiles/pmc_traffic.json
(read by bench.py: `traffic`, `algorithmic_bytes` and their ratio for EVERY kernel of a line, not just the dominant one).

usage: tools/summarize_pmc_all.py TAG DIR [DIR ...]      DIR = rocprofv3 output directories (searched recursively for *counter_collection.csv)

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE under-reports wide coalesced reads 2x on gfx950
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section); the raw (1 x FETCH) figure is kept beside it.  Units per launch come from the
dispatch's grid size and the kernel's lane mapping; algorithmic bytes per unit are the reference's struct sizes in and out
(include/bn254_hip.h: G1 96, G2 192, Gt 384, Fr 32)."""
import collections, csv, glob, json, pathlib, sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
# kernel -> (name in bn254_kernel_stats / bench.py, units per launch from the grid size (threads), algorithmic bytes per unit)
KERNELS = {
    "bn254_miller_naf_B": ("miller", lambda g: g // 2, 96 + 192 + 384),
    "bn254_miller_B": ("miller_reference_schedule", lambda g: g // 2, 96 + 192 + 384),
    "bn254_final_exp_B": ("final_exp", lambda g: g // 2, 384 + 384),
    "bn254_miller_shared2_B": ("miller_shared2", lambda g: g // 2 * 2, 96 + 192 + 384 / 2),
    "bn254_miller_shared4_B": ("miller_shared", lambda g: g // 2 * 4, 96 + 192 + 384 / 4),
    "bn254_miller_prepared_B": ("miller_prepared", lambda g: g // 2, 96 + 384),
    # one SHARED native table (33 792 B per Q, read by every lane): its algorithmic share per pairing is the table once per launch
    "bn254_miller_native_B": ("miller_native", lambda g: g // 2, 96 + 384 + 33792 / 65536),
    # the multi-pairing over native tables, one table per pair: 96 B of P and 33 792 B of table in, a quarter / half of an Fq12 out
    "bn254_miller_native_shared4_B": ("miller_native_shared4", lambda g: g // 2 * 4, 96 + 33792 + 384 / 4),
    "bn254_miller_native_shared2_B": ("miller_native_shared2", lambda g: g // 2 * 2, 96 + 33792 + 384 / 2),
    "bn254_gt_pow_B": ("gt_pow", lambda g: g // 2, 384 + 32 + 384),
    "bn254_gt_mul_B": ("gt_mul", lambda g: g // 2, 3 * 384),
    "bn254_g1_mul_M": ("g1_mul", lambda g: g, 96 + 32 + 96),
    "bn254_g2_mul_M": ("g2_mul", lambda g: g // 2, 192 + 32 + 192),
    "bn254_pairing_W": ("pairing_wave", lambda g: g // 64, 96 + 192 + 384),
    "bn254_final_exp_W": ("final_exp_wave", lambda g: g // 64, 384 + 384),
    "bn254_gt_tail_W": ("gt_tail", lambda g: g // 64, 2 * 384),
    "bn254_gt_reduce_W": ("gt_product", None, 384),                      # units = values in: not visible in the grid (see --values)
    "bn254_miller_naf_Q": ("miller_quad", lambda g: g // 4, 96 + 192 + 384),
    "bn254_final_exp_Q": ("final_exp_quad", lambda g: g // 4, 384 + 384),
}


# the same kernel measured on a SECOND workload gets an entry of its own: passes whose directory name contains the key
WORKLOAD_SUFFIX = {"_pmc_prepq_": ("_per_q", {"bn254_miller_native_B": ("miller_native_per_q", lambda g: g // 2, 96 + 384 + 33792)})}


def load(dirs, counter):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))          # kernel -> grid -> values
    for d in dirs:
        suffix = next((v for k, v in WORKLOAD_SUFFIX.items() if k in str(d)), None)
        for f in glob.glob(str(pathlib.Path(d) / "**" / "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    k = r["Kernel_Name"].split("::")[-1].split("(")[0]
                    if suffix:
                        if k not in suffix[1]:
                            continue                                               # (the other kernels of that pass are measured in their own passes)
                        KERNELS[k + suffix[0]] = suffix[1][k]
                        k = k + suffix[0]
                    agg[k][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, dirs = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("--")]
    values = {a.split("=")[0][2:]: int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("--")}      # --gt_product=65536
    F, W = load(dirs, "FETCH_SIZE"), load(dirs, "WRITE_SIZE")
    lines = [f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, session {tag}; per dispatch, averaged over the dispatches of one grid size.",
             "HBM bytes = (2 x FETCH + WRITE) KB x 1024 (gfx950: FETCH_SIZE counts a 128-byte request as 64: MI355X_MICROARCH.md); raw = (FETCH + WRITE).",
             "algorithmic = the reference's structs in and out (G1 96, G2 192, Gt 384, Fr 32 bytes) per unit x units per launch.", "",
             "%-26s %10s %12s %12s %10s %10s %12s %8s" % ("kernel", "units", "FETCH_KB", "WRITE_KB", "HBM_GB", "raw_GB", "algo_GB", "ratio")]
    out = {}
    for k in sorted(F):
        if k not in KERNELS or k not in W:
            continue
        name, units_of, abytes = KERNELS[k]
        # the largest grid the kernel ran at in this session is the steady-state launch (smaller ones are tails / warm-up shapes)
        grid = max(g for g in F[k] if g in W[k])
        f = sum(F[k][grid]) / len(F[k][grid]); w = sum(W[k][grid]) / len(W[k][grid])
        units = units_of(grid) if units_of else values.get(name)
        hbm = (2 * f + w) * 1024; raw = (f + w) * 1024
        algo = units * abytes if units else None
        lines.append("%-26s %10s %12.1f %12.1f %10.4f %10.4f %12s %8s" % (k, units if units else "?", f, w, hbm / 1e9, raw / 1e9,
                     ("%.4f" % (algo / 1e9)) if algo else "?", ("%.1f" % (hbm / algo)) if algo else "?"))
        if units:
            out[name] = {"kernel": k, "units_per_launch": units, "hbm_bytes_per_launch": hbm, "hbm_bytes_per_launch_raw": raw,
                         "hbm_bytes_per_unit": hbm / units, "algorithmic_bytes_per_unit": abytes, "ratio": hbm / algo,
                         "fetch_kb": f, "write_kb": w, "dispatches": len(F[k][grid]), "source": f"profiles/{tag}_pmc_all.txt"}
    (ROOT / "profiles" / f"{tag}_pmc_all.txt").write_text("\n".join(lines) + "\n")
    tf = ROOT / "profiles" / "pmc_traffic.json"
    old = json.loads(tf.read_text()) if tf.exists() else {}
    old.update(out)                                       # kernels not re-measured in this session keep their earlier entry
    tf.write_text(json.dumps(old, indent=1, sort_keys=True) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

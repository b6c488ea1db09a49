#!/usr/bin/env python3
"""Energy-weighted instruction mix of one kernel of the shipped library, per run of instructions between branches and for a chosen range of
them (a loop): which share of the issue ENERGY is multiply-adds?  The pairing and scalar-multiplication kernels run at the socket's power
cap (bench.py roofline.power_W), so the lever is energy per unit, not issue slots.

usage: tools/energy_mix.py KERNEL_SUBSTRING [--loop FIRST:LAST]     (FIRST / LAST: run numbers as printed, inclusive; default: all runs)

Weights = relative energy per wave instruction, from the all-power-capped 8-waves-per-SIMD issue intervals of profiles/r01h_ubench_valu_rates.txt
(at the cap, time per instruction IS energy per instruction): v_mad_u64_u32 / v_mad_i64_i32 2.66 cycles = 1.00; v_mul_lo / v_mul_hi, 64-bit
shifts and adds, DPP moves, v_alignbit, v_and_or, v_add3, v_cndmask_e64 2.37-2.45 = 0.90; plain 32-bit VOP2 (add, sub, and, xor, shifts)
1.33-1.38 = 0.51; v_mov_b32 1.54 = 0.58; scalar / s_nop / waitcnt: 0.1 (they issue beside the VALU stream); memory instructions 0.9."""
import collections, pathlib, re, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from isa_mix import disassemble, ROOT

def weight(op):
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")): return "mad", 1.00
    if op.startswith(("v_mul_lo", "v_mul_hi")): return "mul32", 0.90
    if op.startswith(("s_", )): return "scalar", 0.10
    if op.startswith(("global_", "flat_", "buffer_", "scratch_", "ds_")): return "memory", 0.90
    if op.startswith("v_mov_b32") and "dpp" in op: return "dpp", 0.90
    if op.startswith(("v_mov", "v_accvgpr")): return "mov", 0.58
    if any(x in op for x in ("b64", "u64", "i64")) or op.startswith(("v_alignbit", "v_and_or", "v_add3", "v_or3", "v_lshl_add", "v_lshl_or", "v_cndmask_b32_e64", "v_bfe", "v_bfi", "v_cmp")): return "valu_wide", 0.90
    if op.startswith("v_"): return "valu32", 0.51
    return "other", 0.10

def runs(text, pat):
    fn = None; cur = collections.Counter(); out = []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            if fn and cur: out.append(cur)
            fn = m.group(1) if pat in m.group(1) else None; cur = collections.Counter(); continue
        if not fn: continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if not m: continue
        cur[m.group(1)] += 1
        if m.group(1).startswith(("s_cbranch", "s_branch", "s_swappc", "s_setpc")):
            out.append(cur); cur = collections.Counter()
    if fn and cur: out.append(cur)
    return out

def summarize(c):
    n = sum(c.values()); cls = collections.Counter(); en = collections.Counter()
    for op, v in c.items():
        k, w = weight(op); cls[k] += v; en[k] += v * w
    tot = sum(en.values())
    return n, cls, en, tot

def main():
    pat = sys.argv[1]
    loop = None
    if "--loop" in sys.argv:
        a, b = sys.argv[sys.argv.index("--loop") + 1].split(":"); loop = (int(a), int(b))
    so = pathlib.Path(__import__("os").environ.get("ISA_LIB", ROOT / "bn_amd" / "libbn254_hip.so"))
    allruns = [r for text in disassemble(so) for r in runs(text, pat)]
    print(f"{pat}: {len(allruns)} runs of instructions between branches / calls; energy weights: mad 1.00, mul32 / 64-bit / dpp / 3-operand / memory 0.90, mov 0.58, 32-bit VOP2 0.51, scalar 0.10")
    print("%4s %6s %7s %7s %7s   %s" % ("run", "instr", "mul%", "E(mul)%", "E(mad)%", "classes"))
    for i, r in enumerate(allruns):
        n, cls, en, tot = summarize(r)
        if n < 60: continue
        mul = cls["mad"] + cls["mul32"]
        print("%4d %6d %6.1f%% %6.1f%% %6.1f%%   %s" % (i, n, 100 * mul / n, 100 * (en["mad"] + en["mul32"]) / tot, 100 * en["mad"] / tot,
                                                   ", ".join(f"{k} {v}" for k, v in cls.most_common())))
    sel = allruns if loop is None else allruns[loop[0]:loop[1] + 1]
    tot_c = collections.Counter()
    for r in sel: tot_c.update(r)
    n, cls, en, tot = summarize(tot_c)
    what = "all runs" if loop is None else f"runs {loop[0]}..{loop[1]}"
    print(f"\n{what}: {n} instructions (static)")
    for k, v in cls.most_common():
        print("   %-10s %6d %5.1f%% of instructions  %5.1f%% of energy" % (k, v, 100 * v / n, 100 * en[k] / tot))
    print("   multiply instructions (mad + mul32): %.1f%% of instructions, %.1f%% of energy" % (100 * (cls["mad"] + cls["mul32"]) / n, 100 * (en["mad"] + en["mul32"]) / tot))
    print("   top opcodes: " + ", ".join(f"{k} {v}" for k, v in tot_c.most_common(14)))

if __name__ == "__main__":
    main()

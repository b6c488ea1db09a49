#!/usr/bin/env python3
"""Bisecting round 4's DPP-fold miscompile: take the GOOD assembly of a unit (combiner off) and fold ONE `v_mov_b32_dpp t, x perm` + `v_sub_u32 d, a, t`
(or `v_sub_u32 d, t, a`) pair by hand into the `v_subrev_u32_dpp d, x, a perm` (`v_sub_u32_dpp d, x, a perm`) the combiner would emit - after checking
that t is dead afterwards.  usage: fold_one_site.py good.s out.s LINE_OF_THE_MOV [--nops N]   (--nops: s_nop N-1 in front of the folded instruction)"""
import re, sys
src, dst, line = sys.argv[1], sys.argv[2], int(sys.argv[3])
nops = int(sys.argv[sys.argv.index("--nops") + 1]) if "--nops" in sys.argv else 0
L = open(src).read().split("\n")
i = line - 1
m = re.match(r"\s*v_mov_b32_dpp (v\d+), (v\d+) (quad_perm:\[[\d,]+\] row_mask:0xf bank_mask:0xf bound_ctrl:1)", L[i])
assert m, L[i]
t, x, perm = m.groups()
s = re.match(r"\s*v_sub_u32_e32 (v\d+), (v\d+), (v\d+)", L[i + 1])
assert s and t in (s.group(2), s.group(3)), L[i + 1]
d, a, b = s.groups()
# t must not be read again before it is written (scan to the end of the function)
for j in range(i + 2, len(L)):
    if L[j].startswith(".Lfunc_end") or "s_endpgm" in L[j]: break
    ops = re.findall(r"\bv\[?(\d+)(?::(\d+))?\]?", L[j].split("//")[0])
    toks = L[j].split()
    if not toks or not toks[0].startswith(("v_", "ds_", "global_", "scratch_", "buffer_")): continue
    tn = int(t[1:])
    regs = [(int(p), int(q) if q else int(p)) for p, q in ops]
    if not regs: continue
    first, rest = regs[0], regs[1:]
    if any(lo <= tn <= hi for lo, hi in rest): raise SystemExit(f"{t} is read again at line {j + 1}: {L[j]}")
    if first[0] <= tn <= first[1] and not toks[0].startswith(("global_store", "scratch_store", "ds_write")): break       # overwritten: dead
folded = (f"\tv_subrev_u32_dpp {d}, {x}, {a} {perm}" if t == b else f"\tv_sub_u32_dpp {d}, {x}, {b} {perm}")
out = L[:i] + ([f"\ts_nop {nops - 1}"] if nops else []) + [folded] + L[i + 2:]
open(dst, "w").write("\n".join(out))
print("folded:", folded.strip(), "(t =", t + ")")

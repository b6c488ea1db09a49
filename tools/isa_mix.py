#!/usr/bin/env python3
"""Static instruction mix of the gfx950 code objects inside libbn254_hip.so (per function): how many mads, other VALU, scratch
(spill) accesses, LDS, calls.  usage: tools/isa_mix.py [substring-of-function-name ...]"""
import collections, pathlib, re, subprocess, sys, tempfile

ROOT = pathlib.Path(__file__).resolve().parents[1]
LLVM = pathlib.Path("/opt/rocm/lib/llvm/bin")

def disassemble(so):
    d = so.read_bytes()
    offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d)]
    out = []
    with tempfile.TemporaryDirectory() as t:
        for i, o in enumerate(offs):
            e = offs[i + 1] if i + 1 < len(offs) else len(d)
            b = pathlib.Path(t) / f"b{i}.bin"; b.write_bytes(d[o:e])
            co = pathlib.Path(t) / f"k{i}.co"
            subprocess.check_call([str(LLVM / "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   f"--input={b}", f"--output={co}", "--unbundle"])
            out.append(subprocess.check_output([str(LLVM / "llvm-objdump"), "-d", str(co)], text=True))
    return out

def classify(op):
    if op.startswith("v_mad_u64_u32"): return "mad64"
    if op.startswith("scratch_load"): return "scratch_ld"
    if op.startswith("scratch_store"): return "scratch_st"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_")): return "global"
    if op.startswith("s_swappc") or op.startswith("s_setpc"): return "call/ret"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "v_mov"
    if op.startswith("v_mul_lo") or op.startswith("v_mul_hi"): return "v_mul32"
    if "b64" in op or "u64" in op or "i64" in op: return "valu64"
    if op.startswith("v_"): return "valu32"
    return "other"

# issue interval seen by ONE wave at 2 waves per SIMD, in cycles (profiles/r01_ubench_valu_rates.txt)
WEIGHT = {"mad64": 9.5, "valu64": 7.5, "v_mul32": 7.5, "valu32": 5.25, "v_mov": 5.25, "salu": 1, "scratch_ld": 5.25, "scratch_st": 5.25,
          "lds": 5.25, "global": 5.25, "waitcnt": 1, "call/ret": 4, "other": 1}

def blocks(text, pat):
    """per basic block (split at branches) of the functions matching `pat`: class counts and a cycle estimate"""
    fn = None; cur = collections.Counter(); first = None
    def flush(tag):
        nonlocal cur, first
        if sum(cur.values()):
            cyc = sum(WEIGHT[k] * v for k, v in cur.items())
            print(f"  {first}..{tag:>9}: n {sum(cur.values()):5d} est_cycles {cyc:8.0f} calls {cur['call/ret']:3d}  " +
                  ", ".join(f"{k} {v}" for k, v in cur.most_common() if k not in ("call/ret",)))
        cur = collections.Counter(); first = None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            if fn: flush("end")
            fn = m.group(1) if pat in m.group(1) else None
            if fn: print(fn[:110])
            continue
        if not fn: continue
        m = re.match(r"^\s+([a-z_0-9]+)\s.*//\s*([0-9A-F]+):", line)
        if not m: continue
        if first is None: first = m.group(2)[-5:]
        cur[classify(m.group(1))] += 1
        if m.group(1).startswith(("s_cbranch", "s_branch")): flush(m.group(2)[-5:])
    if fn: flush("end")

def main():
    pats = sys.argv[1:]
    so = pathlib.Path(__import__("os").environ.get("ISA_LIB", ROOT / "bn_amd" / "libbn254_hip.so"))
    if pats and pats[0] == "--blocks":
        for text in disassemble(so): blocks(text, pats[1])
        return
    for text in disassemble(so):
        fn = None; mix = collections.OrderedDict()
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                fn = m.group(1); mix[fn] = collections.Counter(); continue
            m = re.match(r"^\s+([a-z_0-9]+)\s", line)
            if fn and m:
                mix[fn][classify(m.group(1))] += 1
        for fn, c in mix.items():
            if pats and not any(p in fn for p in pats): continue
            tot = sum(c.values())
            if tot < 20: continue
            print(f"{fn[:90]}\n    total {tot}: " + ", ".join(f"{k} {v}" for k, v in c.most_common()))

if __name__ == "__main__":
    main()

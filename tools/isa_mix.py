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

def main():
    pats = sys.argv[1:]
    for text in disassemble(ROOT / "bn_amd" / "libbn254_hip.so"):
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

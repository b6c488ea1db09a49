#!/usr/bin/env python3
"""Per-kernel resource usage (VGPRs, spills, private segment, LDS) of the gfx950 code objects inside a built library.
usage: tools/kernel_meta.py [path/to/lib.so]      (tests/test_build_quality.py imports kernel_meta())"""
import pathlib, re, subprocess, sys, tempfile
ROOT = pathlib.Path(__file__).resolve().parents[1]
LLVM = pathlib.Path("/opt/rocm/lib/llvm/bin")


def short_name(mangled):
    """bn254_miller_naf_B out of _ZN12_GLOBAL__N_118bn254_miller_naf_BEPKjS1_Pjj"""
    m = re.match(r"^_ZN\d+_GLOBAL__N_1(\d+)", mangled)
    if m:
        n = int(m.group(1)); rest = mangled[m.end():]
        return rest[:n]
    m = re.match(r"^_Z(\d+)", mangled)
    if m:
        n = int(m.group(1)); return mangled[m.end():m.end() + n]
    return mangled


def kernel_meta(so=None):
    """{kernel name: {"vgpr", "spill", "private", "lds", "sgpr"}} of every kernel in the library"""
    so = pathlib.Path(so) if so else ROOT / "bn_amd" / "libbn254_hip.so"
    d = so.read_bytes()
    offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d)]
    out = {}
    with tempfile.TemporaryDirectory() as t:
        for i, o in enumerate(offs):
            e = offs[i + 1] if i + 1 < len(offs) else len(d)
            b = pathlib.Path(t) / f"b{i}.bin"; b.write_bytes(d[o:e])
            co = pathlib.Path(t) / f"k{i}.co"
            subprocess.check_call([str(LLVM / "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={b}", f"--output={co}", "--unbundle"])
            txt = subprocess.check_output([str(LLVM / "llvm-readelf"), "--notes", str(co)], text=True)
            for blk in txt.split("- .agpr_count")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
                out[short_name(g("name"))] = {"vgpr": int(g("vgpr_count")), "spill": int(g("vgpr_spill_count")), "private": int(g("private_segment_fixed_size")),
                                              "lds": int(g("group_segment_fixed_size")), "sgpr": int(g("sgpr_count"))}
    return out


if __name__ == "__main__":
    meta = kernel_meta(sys.argv[1] if len(sys.argv) > 1 else None)
    print("%-34s %5s %6s %8s %7s %7s" % ("kernel", "vgpr", "spill", "private", "lds", "sgpr"))
    for name, m in meta.items():
        print("%-34s %5d %6d %8d %7d %7d" % (name[:34], m["vgpr"], m["spill"], m["private"], m["lds"], m["sgpr"]))

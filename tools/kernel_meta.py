#!/usr/bin/env python3
"""Per-kernel resource usage (VGPRs, spills, private segment, LDS) of the gfx950 code objects inside a built library.
usage: tools/kernel_meta.py [path/to/lib.so]"""
import pathlib, re, subprocess, sys, tempfile
ROOT = pathlib.Path(__file__).resolve().parents[1]
LLVM = pathlib.Path("/opt/rocm/lib/llvm/bin")
so = pathlib.Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "bn_amd" / "libbn254_hip.so"
d = so.read_bytes()
offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", d)]
print("%-34s %5s %6s %8s %7s %7s" % ("kernel", "vgpr", "spill", "private", "lds", "sgpr"))
with tempfile.TemporaryDirectory() as t:
    for i, o in enumerate(offs):
        e = offs[i + 1] if i + 1 < len(offs) else len(d)
        b = pathlib.Path(t) / f"b{i}.bin"; b.write_bytes(d[o:e])
        co = pathlib.Path(t) / f"k{i}.co"
        subprocess.check_call([str(LLVM / "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={b}", f"--output={co}", "--unbundle"])
        txt = subprocess.check_output([str(LLVM / "llvm-readelf"), "--notes", str(co)], text=True)
        for blk in txt.split("- .agpr_count")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            name = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name).rstrip("EPKjS1_PjjiS2_")
            print("%-34s %5s %6s %8s %7s %7s" % (g("name")[:34] if len(name) < 3 else name[:34], g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("sgpr_count")))

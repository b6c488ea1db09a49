#!/usr/bin/env python3
"""Opcode histogram of one function of the gfx950 code objects in a library.  usage: tools/op_hist.py FUNC_SUBSTRING [lib.so]"""
import collections, pathlib, re, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from isa_mix import disassemble, ROOT
pat = sys.argv[1]
so = pathlib.Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "bn_amd" / "libbn254_hip.so"
for text in disassemble(so):
    fn = None; c = collections.Counter()
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            if fn and sum(c.values()) > 20:
                tot = sum(c.values()); print(fn[:100], tot)
                for k, v in c.most_common(40): print(f"   {k:28s} {v:6d} {100*v/tot:5.1f}%")
            fn = m.group(1) if pat in m.group(1) else None; c = collections.Counter(); continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if fn and m: c[m.group(1)] += 1
    if fn and sum(c.values()) > 20:
        tot = sum(c.values()); print(fn[:100], tot)
        for k, v in c.most_common(40): print(f"   {k:28s} {v:6d} {100*v/tot:5.1f}%")

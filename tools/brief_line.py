#!/usr/bin/env python3
"""stdin: output of bench.py; prints value and the per-kernel launch times of the JSON line (or the raw tail if there is none)"""
import json, sys
txt = sys.stdin.read()
for l in txt.splitlines():
    if l.startswith("{"):
        d = json.loads(l); r = d.get("roofline", {})
        print("value %.4g %s  ms/step %.3f  frac %.3f  kernels %s" % (d["value"], d["unit"], d["ms_per_step"], r.get("frac", 0),
              {k: round(v["avg_launch_ms"], 3) for k, v in r.get("kernels", {}).items()}))
        break
else:
    print(txt[-800:])

#!/usr/bin/env python3
"""One-launch product tree (bn254_gt_reduce_W): time over the shape parameters - values per lane pair (`chunk`, lane-pair products,
~20 us each, 32 per wave at once) and lane pairs per wave (`per_wave`, folded one after the other by the wave machine, ~2.7 us
each; fewer per wave = more waves = more levels of the arrival tree, ~5 us each).  Prints one line per size with the best shapes;
the host's policy (product_shape in bn254_hip.hip) is read off this table."""
import json, os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import bn_amd
from bn_amd import distributed as D

dev = torch.device("cuda", 0)
te = D.TorchEngine(bn_amd.Engine(0), dev)

def timed(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

nmax = 1 << 18
P, Q = D.synthetic_points(te, 0, nmax)
f = te.pairing_batch(P, Q)
one = te.empty(48); ref = te.empty(48)
res = {}
for n in (2, 8, 32, 256, 1024, 4096, 1 << 14, 1 << 15, 1 << 16, 1 << 17, 1 << 18):
    for k in ("chunk", "per_wave", "bfly"): te.e.set_option("product_" + k, None)
    te.e.gt_product_dev(f.data_ptr(), n, ref.data_ptr(), te._stream()); torch.cuda.synchronize()
    base = timed(lambda: te.e.gt_product_dev(f.data_ptr(), n, one.data_ptr(), te._stream()))
    rows = []
    for c in (1, 2, 3, 4, 6, 8, 12):
        if c > n: continue
        for L in (2, 4, 8, 16, 32):
            groups = -(-n // c)
            if -(-groups // L) > 16384: continue
            for B in (0, 1, 2, 3):
                if B and L < (2 << B): continue
                te.e.set_option("product_chunk", c); te.e.set_option("product_per_wave", L); te.e.set_option("product_bfly", B)
                t = timed(lambda: te.e.gt_product_dev(f.data_ptr(), n, one.data_ptr(), te._stream()))
                torch.cuda.synchronize()
                assert torch.equal(one, ref), (n, c, L, B)
                rows.append((t, c, L, B))
    rows.sort()
    res[n] = {"default_ms": base, "best": [{"ms": round(t, 4), "chunk": c, "per_wave": L, "bfly": B} for t, c, L, B in rows[:8]]}
    print(n, json.dumps(res[n]), flush=True)

#!/usr/bin/env python3
"""PCIe-inclusive rate of bn254_pairing_batch (pageable numpy buffers) as a function of BN254_OPT_PIPELINE_CHUNK: does cutting ONE machine
round into two half-round chunks on the two pipeline streams hide the copies of a single 2^16 call (VERDICT round 4, weak point 8)?"""
import json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bn_amd
from bn_amd import distributed as D
dev = torch.device("cuda", 0)
te = D.TorchEngine(bn_amd.Engine(0), dev)
nmax = 1 << 18
P, Q = D.synthetic_points(te, 0, nmax)
Pn = P.cpu().numpy().view(np.uint64); Qn = Q.cpu().numpy().view(np.uint64)
e = bn_amd.Engine(0)
out = np.zeros((nmax, 48), np.uint64)
res = {}
for n in (1 << 15, 1 << 16, 1 << 17, 1 << 18):
    row = {}
    for chunk in (None, 1 << 16, 1 << 15, 3 << 13, 1 << 14):
        if chunk is not None and chunk > n: continue
        e.set_option("pipeline_chunk", chunk)
        e.pairing_batch(Pn[:n], Qn[:n], out[:n])
        reps = 6
        t0 = time.perf_counter()
        for _ in range(reps): e.pairing_batch(Pn[:n], Qn[:n], out[:n])
        dt = (time.perf_counter() - t0) / reps
        row["default" if chunk is None else str(chunk)] = {"ms": round(dt * 1e3, 3), "M_per_s": round(n / dt / 1e6, 3)}
    res[str(n)] = row
e.set_option("pipeline_chunk", None)
print(json.dumps(res))

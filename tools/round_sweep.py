#!/usr/bin/env python3
"""2^20 pairings in one call with sub-launches of 2^16 ... 2^20 pairings (BN254_OPT_ROUND_PAIRS): do longer launches help the pairing kernels as they help
the scalar multiplications?  (No: profiles/r06_ab_mul_launch_size.txt.)"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch, bn_amd
from bn_amd import distributed as D
dev = torch.device("cuda", 0)
te = D.TorchEngine(bn_amd.Engine(0), dev); e = te.e
n = 1 << 20
P, Q = D.synthetic_points(te, 0, n)
out = te.empty(n, 48)
ref = None
for rp in (1 << 16, 1 << 17, 1 << 18, 1 << 20, 1 << 16):
    with e.options(round_pairs=rp):
        for _ in range(2): te.pairing_batch(P, Q, out)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(4): te.pairing_batch(P, Q, out)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / 4
    if ref is None: ref = out.clone()
    print(f"round_pairs={rp} {dt*1e3:.2f} ms per 2^20 = {n/dt/1e6:.3f} M pairings/s  same={bool(torch.equal(out, ref))}", flush=True)

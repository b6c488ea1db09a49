#!/usr/bin/env python3
"""ms per call of the normalising G1 scalar multiplication at several call sizes (device-resident).  usage: tools/time_g1mul.py [n ...]"""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import bn_amd
from bn_amd import distributed as D
dev = torch.device("cuda", 0)
te = D.TorchEngine(bn_amd.Engine(0), dev)
sizes = [int(x) for x in sys.argv[1:]] or [1 << 16, 196608, 1 << 18, 1 << 20]
P, Q = D.synthetic_points(te, 0, max(sizes))
k = D.synthetic_scalars_device(te, 1 << 24, (1 << 24) + max(sizes), 1)
for n in sizes:
    q, kk = P[:n].contiguous(), k[:n].contiguous()
    for _ in range(3): te.g1_mul(q, kk)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(10): te.g1_mul(q, kk)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / 10
    print(f"n={n} {dt * 1e3:.3f} ms  {n / dt / 1e6:.2f} M/s")

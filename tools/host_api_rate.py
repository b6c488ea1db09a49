#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point bn254_pairing_batch (pageable numpy buffers in, numpy out).
Never the headline `value`; written to profiles/ for the record."""
import pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bn_amd
from bn_amd import distributed as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
dev = torch.device("cuda", 0)
te = D.TorchEngine(bn_amd.Engine(0), dev)
P, Q = D.synthetic_points(te, 0, n)
Pn = P.cpu().numpy().view(np.uint64); Qn = Q.cpu().numpy().view(np.uint64)
e = bn_amd.Engine(0)
out = np.zeros((n, 48), np.uint64)
e.pairing_batch(Pn, Qn, out)
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    e.pairing_batch(Pn, Qn, out)
dt = (time.perf_counter() - t0) / reps
print(f"bn254_pairing_batch host buffers, n = {n}: {dt*1e3:.2f} ms per call = {n/dt/1e6:.3f} M pairings/s "
      f"(H2D {n*288/1e6:.1f} MB + kernels + D2H {n*384/1e6:.1f} MB, pageable host memory, result buffer reused, chunks of 2^16 on two streams)")

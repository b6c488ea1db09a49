#!/usr/bin/env python3
"""Experiment: does the GPU overlap pairing kernels launched on different streams of one process?  A batch of N pairings is cut
into S equal parts, each on its own stream (own context: own exponentiation table); prints ms per round for several S.
usage: stream_concurrency.py [N]      (run with GPU_MAX_HW_QUEUES=8 in the environment to widen the HW queue pool)"""
import os, pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import bn_amd
from bn_amd import distributed as D

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
dev = torch.device("cuda", 0)
base = D.TorchEngine(bn_amd.Engine(0), dev)
P, Q = D.synthetic_points(base, 0, N)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for S in (1, 2, 4, 8, 16):
    n = N // S
    engs = [D.TorchEngine(bn_amd.Engine(0), dev) for _ in range(S)]
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    Ps = [P[i * n:(i + 1) * n].contiguous() for i in range(S)]; Qs = [Q[i * n:(i + 1) * n].contiguous() for i in range(S)]
    outs = [base.empty(n, 48) for _ in range(S)]
    def run(k):
        for _ in range(k):
            for s in range(S):
                with torch.cuda.stream(streams[s]):
                    engs[s].pairing_batch(Ps[s], Qs[s], outs[s])
    run(2); torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); run(10); torch.cuda.synchronize(dev); dt = (time.perf_counter() - t0) / 10
    print(f"S = {S:2d} streams x {n:6d} pairings: {dt * 1e3:7.3f} ms per round = {N / dt / 1e6:.3f} M pairings/s")

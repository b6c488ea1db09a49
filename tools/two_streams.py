#!/usr/bin/env python3
"""Experiment: aggregate pairing throughput of S independent streams (one context each) on ONE GPU, each running full batches or
fractions of the batch; shows how much the kernel-boundary tails cost.  usage: two_streams.py [streams] [batch_per_stream]"""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import bn_amd
from bn_amd import distributed as D

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
steps = 30
dev = torch.device("cuda", 0)
engs = [D.TorchEngine(bn_amd.Engine(0), dev) for _ in range(S)]
streams = [torch.cuda.Stream(dev) for _ in range(S)]
P, Q = D.synthetic_points(engs[0], 0, n)
outs = [engs[0].empty(n, 48) for _ in range(S)]
def run(k):
    for _ in range(k):
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                engs[s].pairing_batch(P, Q, outs[s])
run(3); torch.cuda.synchronize(dev)
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(dev); dt = time.perf_counter() - t0
print(f"streams {S} batch/stream {n}: {S * n * steps / dt / 1e6:.3f} M pairings/s, {dt / steps * 1e3:.3f} ms per round")
assert all(torch.equal(o, outs[0]) for o in outs)

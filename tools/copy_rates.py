#!/usr/bin/env python3
"""Experiment: host<->device copy rates that bound the host-buffer API (pageable vs pinned, host memcpy into pinned staging)."""
import time, numpy as np, torch
dev = torch.device("cuda", 0)
for mb in (19, 25, 64):
    n = mb * (1 << 20) // 8
    pageable = torch.empty(n, dtype=torch.int64); pageable.random_()
    pinned = torch.empty(n, dtype=torch.int64).pin_memory()
    d = torch.empty(n, dtype=torch.int64, device=dev)
    def t(f, reps=5):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    a = t(lambda: d.copy_(pageable)); b = t(lambda: d.copy_(pinned, non_blocking=True))
    c = t(lambda: pageable.copy_(d)); e = t(lambda: pinned.copy_(d, non_blocking=True))
    f = t(lambda: pinned.copy_(pageable))
    src = pageable.numpy(); dst = pinned.numpy()
    g = t(lambda: np.copyto(dst, src))
    print(f"{mb} MB: H2D pageable {mb/a/1e3:.1f} GB/s ({a*1e3:.2f} ms), H2D pinned {mb/b/1e3:.1f} GB/s ({b*1e3:.2f} ms), D2H pageable {mb/c/1e3:.1f} GB/s ({c*1e3:.2f} ms), "
          f"D2H pinned {mb/e/1e3:.1f} GB/s ({e*1e3:.2f} ms), host memcpy pageable->pinned torch {mb/f/1e3:.1f} GB/s ({f*1e3:.2f} ms) numpy 1 thread {mb/g/1e3:.1f} GB/s ({g*1e3:.2f} ms)")

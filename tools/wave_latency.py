#!/usr/bin/env python3
"""Latency of the small-batch / tail paths on the GPU box: final exponentiation per wave vs per lane pair, the one-launch product
tree, the product-then-exponentiate tail, and a by-value pairing(p, q) through the host API.  Prints one JSON object."""
import json, os, sys, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bn_amd
from bn_amd import distributed as D

dev = torch.device("cuda", 0)
te = D.TorchEngine(bn_amd.Engine(0), dev)

def timed(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

res = {"final_exp_ms": {}, "product_tree_ms": {}, "tail_ms": {}}
nmax = 1 << 18
P, Q = D.synthetic_points(te, 0, nmax)
f = te.empty(nmax, 48)
te.e.miller_batch_dev(P.data_ptr(), Q.data_ptr(), f.data_ptr(), nmax, te._stream())
out = te.empty(nmax, 48)
for n in (1, 8, 64, 256, 1024, 2048, 4096, 6144, 8192):
    row = {}
    for name, thr in (("wave", 1 << 20), ("lane_pair", 0)):
        te.e.set_option("wave_fe_max", thr)
        row[name] = timed(lambda: te.e.final_exp_batch_dev(f.data_ptr(), out.data_ptr(), n, te._stream()), reps=10)
    res["final_exp_ms"][n] = row
te.e.set_option("wave_fe_max", None)
one = te.empty(48)
for n in (32, 1024, 1 << 15, 1 << 16, 1 << 18):
    res["product_tree_ms"][n] = timed(lambda: te.e.gt_product_dev(f.data_ptr(), n, one.data_ptr(), te._stream()), reps=10)
for m in (1, 2, 8, 64):
    res["tail_ms"][m] = timed(lambda: te.e.gt_product_final_exp_dev(f.data_ptr(), m, one.data_ptr(), te._stream()), reps=10)
# whole multi-pairing on one GPU: 2^15 and 2^18 pairs (BASELINE configs[3] per-GPU shard and total)
res["pairing_product_ms"] = {n: timed(lambda: D.pairing_product_sharded(te, P[:n], Q[:n]), reps=5, warm=2) for n in (1, 4, 1 << 15, 1 << 18)}
for name, thr in (("wave", 1 << 20), ("lane_pair", 0)):
    te.e.set_option("wave_pairing_max", thr); te.e.set_option("wave_fe_max", 1024 if thr else 0); te.e.set_option("quad_max", 0)
    res.setdefault("pairing_batch_ms", {})[name] = {n: timed(lambda: te.e.pairing_batch_dev(P.data_ptr(), Q.data_ptr(), out.data_ptr(), n, te._stream()), reps=5, warm=1) for n in (1, 4, 64, 256, 768, 1024, 2048, 3072, 4096, 6144)}
for k in ("wave_pairing_max", "wave_fe_max", "quad_max"): te.e.set_option(k, None)
# the mid-size window: four lanes per pairing (bn254_kernels_q.hip) against the lane-pair kernels and the library's own choice
mid = {}
for name, opts in (("default", {}), ("quad", {"wave_pairing_max": 0, "wave_fe_max": 0, "quad_max": 1 << 20}), ("lane_pair", {"wave_pairing_max": 0, "wave_fe_max": 0, "quad_max": 0}),
                   ("wave", {"wave_pairing_max": 1 << 20, "wave_fe_max": 1 << 20})):
    for k, v in opts.items(): te.e.set_option(k, v)
    mid[name] = {n: timed(lambda: te.e.pairing_batch_dev(P.data_ptr(), Q.data_ptr(), out.data_ptr(), n, te._stream()), reps=5, warm=1)
                 for n in (1024, 4096, 5120, 5121, 6144, 8192, 12288, 16384, 16385, 24576, 32768, 65536) if not (name == "wave" and n > 8192)}
    for k in opts: te.e.set_option(k, None)
res["pairing_batch_ms_mid_size"] = mid
res["miller_only_ms"] = {n: timed(lambda: te.e.miller_batch_dev(P.data_ptr(), Q.data_ptr(), f.data_ptr(), n, te._stream()), reps=5, warm=1) for n in (1, 1 << 15, 1 << 16)}
# by-value pairing through the host-buffer API (what `pairing(p, q)` of lib.rs:181-183 costs a caller)
e = bn_amd.Engine(0)
Pn = P[:64].cpu().numpy().view(np.uint64); Qn = Q[:64].cpu().numpy().view(np.uint64)
outn = np.zeros((64, 48), np.uint64)
for n in (1, 4, 64):
    e.pairing_batch(Pn[:n], Qn[:n], outn[:n])
    t0 = time.perf_counter()
    for _ in range(20): e.pairing_batch(Pn[:n], Qn[:n], outn[:n])
    res.setdefault("host_pairing_batch_ms", {})[n] = (time.perf_counter() - t0) / 20 * 1e3
res["wave_program_us"] = {name: e.wave_ubench(k, 50 if k == 4 else 400) for k, name in enumerate(("cyclotomic_sqr_2_phases", "product_3_phases", "slot_copy_1_comb", "frobenius_1_prod", "final_exp", "five_fused_squarings_6_phases"))}
print(json.dumps(res))

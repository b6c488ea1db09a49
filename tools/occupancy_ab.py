#!/usr/bin/env python3
"""A real kernel at two and at three resident waves per SIMD: bn254_exp_by_neg_z_B - the reference's
exponentiation by u, 62 Granger-Scott squarings + 27 dense products on one Fq12 in registers - compiles to 4 spilled VGPRs at 168 VGPRs
as at 256, so the SAME instruction stream can be timed at both occupancies; bn254_gt_mul_B (one dense product) likewise.
Run once per library (BN254_LIB_PATH): the default build and build_variants/lib_w3.so (bn254_kernels_b.hip with -DBN_WAVES=3).
Prints one JSON line: ms per launch and elements per second for n = 1, 2, 3, 4, 6 waves per SIMD worth of elements."""
import json, os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import bn_amd
from bn_amd import distributed as D

dev = torch.device("cuda", 0)
te = D.TorchEngine(bn_amd.Engine(0), dev)
cus = torch.cuda.get_device_properties(0).multi_processor_count
per_wave_slot = cus * 4 * 32                 # elements that put ONE wave on every SIMD (a lane pair per element)
nmax = 6 * per_wave_slot
P, Q = D.synthetic_points(te, 0, 1 << 16)
f16 = te.empty(1 << 16, 48)
te.e.miller_batch_dev(P.data_ptr(), Q.data_ptr(), f16.data_ptr(), 1 << 16, te._stream())
f = f16.repeat((nmax + (1 << 16) - 1) // (1 << 16), 1)[:nmax].contiguous()
g = f.flip(0).contiguous()
out = te.empty(nmax, 48)

def timed(fn, reps=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

res = {"lib": os.environ.get("BN254_LIB_PATH", "default"), "cus": cus, "exp_by_neg_z": {}, "gt_mul": {}}
for w in (1, 2, 3, 4, 6):
    n = w * per_wave_slot
    ms = timed(lambda: te.e.exp_by_neg_z_dev(f.data_ptr(), out.data_ptr(), n, te._stream()))
    res["exp_by_neg_z"][w] = {"n": n, "ms": round(ms, 4), "M_per_s": round(n / ms / 1e3, 3)}
    ms = timed(lambda: te.e.gt_mul_dev(f.data_ptr(), g.data_ptr(), out.data_ptr(), n, te._stream()), reps=20)
    res["gt_mul"][w] = {"n": n, "ms": round(ms, 4), "M_per_s": round(n / ms / 1e3, 3)}
print(json.dumps(res))

#!/usr/bin/env python3
"""Where do the 7.9 ms of ONE bn254_pairing_batch(2^16) call on pageable host buffers go (6.5 ms of kernels + 0.85 ms of link-rate copies
+ ?) - VERDICT round 5, item 7.  Replays what csrc/bn254_multi.hip run_slot_chunks does for a one-chunk call - H2D p, H2D q, Miller loop,
final exponentiation, D2H, stream synchronise - on ONE stream through the same HIP runtime, with a host time stamp around every API call
(a copy from / to pageable memory blocks the host while the runtime stages it) and HIP events between the phases (what the GPU sees),
for pageable and for pinned host buffers; then times the real entry point for comparison.
Writes a table; run on the GPU box."""
import ctypes as C
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import bn_amd
from bn_amd import _native
from bn_amd import distributed as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
dev = torch.device("cuda", 0)
eng = bn_amd.Engine(0)
te = D.TorchEngine(eng, dev)
P, Q = D.synthetic_points(te, 0, n)
Pn = P.cpu().numpy().view(np.uint64).copy(); Qn = Q.cpu().numpy().view(np.uint64).copy()
out = np.zeros((n, 48), np.uint64)
hip = C.CDLL(_native._preload_shared_hip_runtime() or "libamdhip64.so")
H2D, D2H = 1, 2
stream = torch.cuda.Stream(dev)
s = C.c_void_p(stream.cuda_stream)
dp = te.empty(n, 12); dq = te.empty(n, 24); dout = te.empty(n, 48)


def ck(rc):
    assert rc == 0, rc


def events(k):
    ev = []
    for _ in range(k):
        e = C.c_void_p(); ck(hip.hipEventCreate(C.byref(e))); ev.append(e)
    return ev


def one_call(hp, hq, ho):
    """returns (host stamps in ms relative to the start, GPU phase durations in ms)"""
    ev = events(6)
    st = [time.perf_counter()]
    ck(hip.hipEventRecord(ev[0], s))
    ck(hip.hipMemcpyAsync(C.c_void_p(dp.data_ptr()), C.c_void_p(hp), C.c_size_t(n * 96), H2D, s)); st.append(time.perf_counter())
    ck(hip.hipEventRecord(ev[1], s))
    ck(hip.hipMemcpyAsync(C.c_void_p(dq.data_ptr()), C.c_void_p(hq), C.c_size_t(n * 192), H2D, s)); st.append(time.perf_counter())
    ck(hip.hipEventRecord(ev[2], s))
    eng.pairing_batch_dev(dp.data_ptr(), dq.data_ptr(), dout.data_ptr(), n, stream.cuda_stream); st.append(time.perf_counter())
    ck(hip.hipEventRecord(ev[3], s))
    ck(hip.hipMemcpyAsync(C.c_void_p(ho), C.c_void_p(dout.data_ptr()), C.c_size_t(n * 384), D2H, s)); st.append(time.perf_counter())
    ck(hip.hipEventRecord(ev[4], s))
    ck(hip.hipStreamSynchronize(s)); st.append(time.perf_counter())
    ms = C.c_float()
    gpu = []
    for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (0, 4)):
        ck(hip.hipEventElapsedTime(C.byref(ms), ev[a], ev[b])); gpu.append(ms.value)
    for e in ev:
        hip.hipEventDestroy(e)
    return [(x - st[0]) * 1e3 for x in st[1:]], gpu


def table(name, hp, hq, ho, reps=6):
    one_call(hp, hq, ho)
    rows = [one_call(hp, hq, ho) for _ in range(reps)]
    host = np.median(np.array([r[0] for r in rows]), axis=0); gpu = np.median(np.array([r[1] for r in rows]), axis=0)
    print(f"--- {name}: n = {n}, median of {reps} calls")
    print("  host thread returns from:  H2D p %.3f   H2D q %.3f   launches %.3f   D2H %.3f   stream sync %.3f ms  (cumulative, since the call began)" % tuple(host))
    print("  on the stream (HIP events): H2D p %.3f   H2D q %.3f   kernels %.3f   D2H %.3f   first event -> last %.3f ms" % tuple(gpu))
    print("  bytes: H2D %.1f MB at %.1f GB/s, D2H %.1f MB at %.1f GB/s" % (n * 288 / 1e6, n * 288 / 1e6 / (gpu[0] + gpu[1]), n * 384 / 1e6, n * 384 / 1e6 / gpu[3]))
    return host, gpu


print("one bn254_pairing_batch call replayed phase by phase on one stream (tools/host_call_timeline.py)")
hp_, gp_ = table("pageable numpy buffers", Pn.ctypes.data, Qn.ctypes.data, out.ctypes.data)
# the same with pinned buffers (what a caller that controls its allocations can do: hipHostMalloc / hipHostRegister)
pin = []
for a in (Pn, Qn, out):
    t = torch.empty(a.shape, dtype=torch.int64).pin_memory()
    t.numpy().view(np.uint64)[:] = a
    pin.append(t)
hp2, gp2 = table("pinned host buffers", pin[0].data_ptr(), pin[1].data_ptr(), pin[2].data_ptr())
# registering the CALLER's pageable buffers on the fly: what would it cost per call?
t0 = time.perf_counter()
for a in (Pn, Qn, out):
    ck(hip.hipHostRegister(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 0))
t1 = time.perf_counter()
hp3, gp3 = table("pageable buffers after hipHostRegister", Pn.ctypes.data, Qn.ctypes.data, out.ctypes.data)
t2 = time.perf_counter()
for a in (Pn, Qn, out):
    ck(hip.hipHostUnregister(C.c_void_p(a.ctypes.data)))
t3 = time.perf_counter()
print("  hipHostRegister of the three buffers: %.3f ms, hipHostUnregister: %.3f ms (per call, if done inside the entry point)" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3))
# the real entry point
e2 = bn_amd.Engine(0)
e2.pairing_batch(Pn, Qn, out)
ts = []
for _ in range(8):
    t0 = time.perf_counter(); e2.pairing_batch(Pn, Qn, out); ts.append((time.perf_counter() - t0) * 1e3)
print("--- bn254_pairing_batch itself (pageable numpy buffers): median %.3f ms per call, min %.3f" % (float(np.median(ts)), min(ts)))
pt = [t.numpy().view(np.uint64) for t in pin]
e2.pairing_batch(pt[0], pt[1], pt[2])
ts = []
for _ in range(8):
    t0 = time.perf_counter(); e2.pairing_batch(pt[0], pt[1], pt[2]); ts.append((time.perf_counter() - t0) * 1e3)
print("--- bn254_pairing_batch itself on PINNED buffers: median %.3f ms per call, min %.3f" % (float(np.median(ts)), min(ts)))


# why are the kernels 6.9 ms inside a call and 6.5 ms in bench.py's back-to-back steps?  The same two launches (a) back to back, (b) each
# pair preceded by an idle gap of the length of the copies: the GPU's clocks / power state after the gap
def kernels_only(gap_ms, reps=8):
    ev = events(2); ms = C.c_float(); res = []
    for _ in range(reps + 2):
        if gap_ms:
            ck(hip.hipStreamSynchronize(s)); time.sleep(gap_ms * 1e-3)
        ck(hip.hipEventRecord(ev[0], s))
        eng.pairing_batch_dev(dp.data_ptr(), dq.data_ptr(), dout.data_ptr(), n, stream.cuda_stream)
        ck(hip.hipEventRecord(ev[1], s))
        if gap_ms:
            ck(hip.hipStreamSynchronize(s)); ck(hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1])); res.append(ms.value)
    if not gap_ms:
        ck(hip.hipStreamSynchronize(s)); ck(hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1])); res.append(ms.value)
    return float(np.median(res[2:] if gap_ms else res))


print("--- the two kernels alone (device-resident inputs): last of 10 back-to-back pairs %.3f ms; after an idle gap of 0.4 ms %.3f ms, 1 ms %.3f ms, 5 ms %.3f ms"
      % (kernels_only(0), kernels_only(0.4), kernels_only(1.0), kernels_only(5.0)))

#!/usr/bin/env python3
"""Experiment (needs a -DBN_STAMP build via BN254_LIB_PATH): distribution of per-wave start/end times of the Miller kernel."""
import ctypes as C, pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, torch
import bn_amd
from bn_amd import _native, distributed as D
dev = torch.device("cuda", 0)
eng = D.TorchEngine(bn_amd.Engine(0), dev)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
P, Q = D.synthetic_points(eng, 0, n)
out = eng.empty(n, 48)
which = sys.argv[1] if len(sys.argv) > 1 else "miller"
for _ in range(5):
    eng.e.miller_batch_dev(P.data_ptr(), Q.data_ptr(), out.data_ptr(), n, eng._stream())
torch.cuda.synchronize()
if which == "final_exp":                       # the stamps of the LAST kernel that ran are the ones read back
    out2 = eng.empty(n, 48)
    for _ in range(5):
        eng.e.final_exp_batch_dev(out.data_ptr(), out2.data_ptr(), n, eng._stream())
    torch.cuda.synchronize()
lib = _native.lib()
nw = (2 * n + 63) // 64
st = np.zeros(3 * nw, np.uint64)
lib.bn254_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.bn254_debug_stamps(st.ctypes.data, st.size) == 0
st = st.reshape(nw, 3)
t0, t1, xcc = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64), st[:, 2]
base = t0.min()
print("kernel span (100 MHz ticks -> us):", (t1.max() - base) / 100)
print("start  us: min %.1f  p50 %.1f  p99 %.1f  max %.1f" % tuple(np.percentile((t0 - base) / 100, [0, 50, 99, 100])))
print("end    us: min %.1f  p01 %.1f  p50 %.1f  p99 %.1f  max %.1f" % tuple(np.percentile((t1 - base) / 100, [0, 1, 50, 99, 100])))
print("life   us: min %.1f  p50 %.1f  max %.1f   mean/span = %.3f" % (*np.percentile((t1 - t0) / 100, [0, 50, 100]), (t1 - t0).mean() / (t1.max() - base)))
hw = (st[:, 2] & np.uint64(0xffffffff)).astype(np.int64); xcc = (st[:, 2] >> np.uint64(32)).astype(np.int64)
wave_id = hw & 15; simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
import collections
pairs = collections.defaultdict(list)
for i in range(nw): pairs[int(key[i])].append(((t1[i] - t0[i]) / 100, int(wave_id[i])))
sizes = collections.Counter(len(v) for v in pairs.values()); print("waves per SIMD histogram:", dict(sizes), " distinct SIMDs:", len(pairs))
two = [sorted(v) for v in pairs.values() if len(v) == 2]
if two:
    a = np.array([[v[0][0], v[1][0]] for v in two]); print("pairs: first-finisher mean %.1f us, second mean %.1f us; sum mean %.1f" % (a[:, 0].mean(), a[:, 1].mean(), a.sum(1).mean()))
    print("  first-finisher wave_id counts:", collections.Counter(v[0][1] for v in two))
hist, edges = np.histogram((t1 - t0) / 100, bins=12); print("lifetime histogram:", list(zip(edges[:-1].astype(int).tolist(), hist.tolist())))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("  xcc %d: waves %4d  mean life %.1f us  end p50 %.1f max %.1f" % (x, m.sum(), (t1 - t0)[m].mean() / 100, np.median((t1 - base)[m]) / 100, (t1 - base)[m].max() / 100))

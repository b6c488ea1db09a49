"""TEST INFRASTRUCTURE - ctypes binding of oracle/libbn_oracle.so (the reference-faithful C restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package bn_amd/ never does.  All values are numpy uint64 arrays in the reference's #[repr(C)] layouts
(Montgomery limbs): Fq/Fr (4,), Fq2 (8,), Fq12/Gt (48,), G1 (12,), G2 (24,); batches add a leading axis.
"""
import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
FQ, FR = 0, 1
_U64P = C.POINTER(C.c_uint64)


def build(force=False):
    """compile the oracle with gcc (idempotent)."""
    libs = [HERE / "libbn_oracle.so", HERE / "libbn_oracle_native128.so"]
    src_m = max((HERE / f).stat().st_mtime for f in ("bn_oracle.c", "bn_model.py", "gen_consts.py", "Makefile"))
    if force or any((not l.exists()) or l.stat().st_mtime < src_m for l in libs):
        subprocess.check_call(["make", "-s", "-C", str(HERE), "CC=gcc"])
    return libs


def _load(native128=False):
    name = "libbn_oracle_native128.so" if native128 else "libbn_oracle.so"
    path = HERE / name
    if not path.exists():
        build()
    return C.CDLL(str(path))


def _p(a):
    return a.ctypes.data_as(_U64P)


def usable_cpus():
    """hardware threads this process may actually use: the affinity mask, capped by a cgroup v2 CPU quota (the GPU box
    shows 256 threads but grants 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _u64(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if n is not None:
        assert a.size == n, (a.shape, n)
    return a


class Oracle:
    def __init__(self, native128=False):
        self.lib = _load(native128)
        self.lib.bno_pairing_batch.argtypes = [_U64P, _U64P, _U64P, C.c_size_t, C.c_int]
        for f in ("bno_g1_mul_batch", "bno_g2_mul_batch", "bno_g1_mul_batch_jacobian", "bno_g2_mul_batch_jacobian"):
            getattr(self.lib, f).argtypes = [_U64P, _U64P, _U64P, C.c_size_t, C.c_int]
        self.lib.bno_pairing_product.argtypes = [_U64P, _U64P, C.c_size_t, _U64P]
        self.lib.bno_fp_from_decimal.argtypes = [C.c_int, C.c_char_p, _U64P]
        self.native128 = bool(self.lib.bno_native128())

    # ---- generic call helpers
    def _un(self, fn, a, n_in, n_out, *extra):
        a = _u64(a, n_in); o = np.empty(n_out, np.uint64)
        getattr(self.lib, fn)(_p(a), *extra, _p(o)); return o

    def _bin(self, fn, a, b, n_a, n_b, n_out):
        a = _u64(a, n_a); b = _u64(b, n_b); o = np.empty(n_out, np.uint64)
        getattr(self.lib, fn)(_p(a), _p(b), _p(o)); return o

    # ---- Fq / Fr
    def fp_from_int(self, w, v):
        raw = np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64); o = np.empty(4, np.uint64)
        if not self.lib.bno_fp_from_raw(w, _p(raw), _p(o)):
            raise ValueError("integer is not less than modulus")
        return o

    def fp_to_int(self, w, a):
        a = _u64(a, 4); o = np.empty(4, np.uint64); self.lib.bno_fp_to_raw(w, _p(a), _p(o))
        return sum(int(x) << (64 * i) for i, x in enumerate(o))

    def fp_from_decimal(self, w, s):
        o = np.empty(4, np.uint64)
        if not self.lib.bno_fp_from_decimal(w, s.encode(), _p(o)):
            return None
        return o

    def fp_add(self, w, a, b): o = np.empty(4, np.uint64); self.lib.bno_fp_add(w, _p(_u64(a, 4)), _p(_u64(b, 4)), _p(o)); return o
    def fp_sub(self, w, a, b): o = np.empty(4, np.uint64); self.lib.bno_fp_sub(w, _p(_u64(a, 4)), _p(_u64(b, 4)), _p(o)); return o
    def fp_mul(self, w, a, b): o = np.empty(4, np.uint64); self.lib.bno_fp_mul(w, _p(_u64(a, 4)), _p(_u64(b, 4)), _p(o)); return o
    def fp_neg(self, w, a): o = np.empty(4, np.uint64); self.lib.bno_fp_neg(w, _p(_u64(a, 4)), _p(o)); return o
    def fp_inverse(self, w, a):
        o = np.empty(4, np.uint64)
        return o if self.lib.bno_fp_inverse(w, _p(_u64(a, 4)), _p(o)) else None

    # ---- tower
    def fq2_mul(self, a, b): return self._bin("bno_fq2_mul", a, b, 8, 8, 8)
    def fq2_sqr(self, a): return self._un("bno_fq2_sqr", a, 8, 8)
    def fq2_inverse(self, a): return self._un("bno_fq2_inverse", a, 8, 8)
    def fq2_mul_xi(self, a): return self._un("bno_fq2_mul_xi", a, 8, 8)
    def fq6_mul(self, a, b): return self._bin("bno_fq6_mul", a, b, 24, 24, 24)
    def fq6_sqr(self, a): return self._un("bno_fq6_sqr", a, 24, 24)
    def fq6_inverse(self, a): return self._un("bno_fq6_inverse", a, 24, 24)
    def fq12_one(self): o = np.empty(48, np.uint64); self.lib.bno_fq12_one(_p(o)); return o
    def fq12_mul(self, a, b): return self._bin("bno_fq12_mul", a, b, 48, 48, 48)
    def fq12_add(self, a, b): return self._bin("bno_fq12_add", a, b, 48, 48, 48)
    def fq12_sub(self, a, b): return self._bin("bno_fq12_sub", a, b, 48, 48, 48)
    def fq12_sqr(self, a): return self._un("bno_fq12_sqr", a, 48, 48)
    def fq12_neg(self, a): return self._un("bno_fq12_neg", a, 48, 48)
    def fq12_inverse(self, a): return self._un("bno_fq12_inverse", a, 48, 48)
    def fq12_unitary_inverse(self, a): return self._un("bno_fq12_unitary_inverse", a, 48, 48)
    def fq12_frobenius_map(self, a, p): return self._un("bno_fq12_frobenius_map", a, 48, 48, C.c_int(p))
    def fq12_cyclotomic_squared(self, a): return self._un("bno_fq12_cyclotomic_squared", a, 48, 48)
    def fq12_exp_by_neg_z(self, a): return self._un("bno_fq12_exp_by_neg_z", a, 48, 48)
    def fq12_final_exponentiation(self, a): return self._un("bno_fq12_final_exponentiation", a, 48, 48)
    def fq12_final_exp_first_chunk(self, a): return self._un("bno_fq12_final_exp_first_chunk", a, 48, 48)
    def fq12_mul_by_024(self, f, l0, lvw, lvv):
        o = np.empty(48, np.uint64)
        self.lib.bno_fq12_mul_by_024(_p(_u64(f, 48)), _p(_u64(l0, 8)), _p(_u64(lvw, 8)), _p(_u64(lvv, 8)), _p(o)); return o
    def gt_pow(self, a, fr): return self._bin("bno_gt_pow", a, fr, 48, 4, 48)

    # ---- groups
    def g1_one(self): o = np.empty(12, np.uint64); self.lib.bno_g1_one(_p(o)); return o
    def g2_one(self): o = np.empty(24, np.uint64); self.lib.bno_g2_one(_p(o)); return o
    def g1_zero(self): o = np.empty(12, np.uint64); self.lib.bno_g1_zero(_p(o)); return o
    def g2_zero(self): o = np.empty(24, np.uint64); self.lib.bno_g2_zero(_p(o)); return o
    def g1_add(self, a, b): return self._bin("bno_g1_add", a, b, 12, 12, 12)
    def g2_add(self, a, b): return self._bin("bno_g2_add", a, b, 24, 24, 24)
    def g1_double(self, a): return self._un("bno_g1_double", a, 12, 12)
    def g2_double(self, a): return self._un("bno_g2_double", a, 24, 24)
    def g1_neg(self, a): return self._un("bno_g1_neg", a, 12, 12)
    def g2_neg(self, a): return self._un("bno_g2_neg", a, 24, 24)
    def g1_mul(self, a, fr): return self._bin("bno_g1_mul", a, fr, 12, 4, 12)
    def g2_mul(self, a, fr): return self._bin("bno_g2_mul", a, fr, 24, 4, 24)
    def g1_normalize(self, a): return self._un("bno_g1_normalize", a, 12, 12)
    def g2_normalize(self, a): return self._un("bno_g2_normalize", a, 24, 24)
    def g1_eq(self, a, b): return bool(self.lib.bno_g1_eq(_p(_u64(a, 12)), _p(_u64(b, 12))))
    def g2_eq(self, a, b): return bool(self.lib.bno_g2_eq(_p(_u64(a, 24)), _p(_u64(b, 24))))
    def g1_to_affine(self, a):
        o = np.empty(8, np.uint64)
        return o if self.lib.bno_g1_to_affine(_p(_u64(a, 12)), _p(o)) else None
    def g2_to_affine(self, a):
        o = np.empty(16, np.uint64)
        return o if self.lib.bno_g2_to_affine(_p(_u64(a, 24)), _p(o)) else None

    # ---- pairing
    def g2_precompute(self, q_aff):
        o = np.empty(102 * 24, np.uint64)
        n = self.lib.bno_g2_precompute(_p(_u64(q_aff, 16)), _p(o)); assert n == 102
        return o.reshape(102, 3, 8)          # [i][ell_0 | ell_vw | ell_vv][Fq2]
    def miller_loop(self, coeffs, p_aff): return self._bin("bno_miller_loop", coeffs, p_aff, 102 * 24, 8, 48)
    def miller_only(self, p, q): return self._bin("bno_miller_only", p, q, 12, 24, 48)
    def pairing(self, p, q): return self._bin("bno_pairing", p, q, 12, 24, 48)

    def pairing_batch(self, p, q, nthreads=None):
        p = _u64(p).reshape(-1, 12); q = _u64(q).reshape(-1, 24); n = p.shape[0]; assert q.shape[0] == n
        o = np.empty((n, 48), np.uint64)
        self.lib.bno_pairing_batch(_p(p), _p(q), _p(o), n, nthreads or usable_cpus()); return o

    def pairing_product(self, p, q):
        p = _u64(p).reshape(-1, 12); q = _u64(q).reshape(-1, 24); n = p.shape[0]; assert q.shape[0] == n
        o = np.empty(48, np.uint64); self.lib.bno_pairing_product(_p(p), _p(q), n, _p(o)); return o

    def _mulb(self, fn, w, p, k, nthreads):
        p = _u64(p).reshape(-1, w); k = _u64(k).reshape(-1, 4); n = p.shape[0]; assert k.shape[0] == n
        o = np.empty((n, w), np.uint64)
        getattr(self.lib, fn)(_p(p), _p(k), _p(o), n, nthreads or usable_cpus()); return o
    def g1_mul_batch(self, p, k, nthreads=None): return self._mulb("bno_g1_mul_batch", 12, p, k, nthreads)
    def g2_mul_batch(self, p, k, nthreads=None): return self._mulb("bno_g2_mul_batch", 24, p, k, nthreads)
    def g1_mul_batch_jacobian(self, p, k, nthreads=None): return self._mulb("bno_g1_mul_batch_jacobian", 12, p, k, nthreads)
    def g2_mul_batch_jacobian(self, p, k, nthreads=None): return self._mulb("bno_g2_mul_batch_jacobian", 24, p, k, nthreads)

    # ---- wire format (fixed-size batch records: G1 65 bytes, G2 129 bytes)
    def fr_encode(self, k):
        o = np.zeros(32, np.uint8); self.lib.bno_fr_encode(_p(_u64(k, 4)), o.ctypes.data_as(C.c_void_p)); return o
    def fr_decode(self, b):
        b = np.ascontiguousarray(b, np.uint8); o = np.zeros(4, np.uint64)
        return int(self.lib.bno_fr_decode(b.ctypes.data_as(C.c_void_p), _p(o))), o
    def g1_encode(self, p):
        o = np.zeros(65, np.uint8); self.lib.bno_g1_encode(_p(_u64(p, 12)), o.ctypes.data_as(C.c_void_p)); return o
    def g2_encode(self, p):
        o = np.zeros(129, np.uint8); self.lib.bno_g2_encode(_p(_u64(p, 24)), o.ctypes.data_as(C.c_void_p)); return o
    def g1_decode(self, b):
        b = np.ascontiguousarray(b, np.uint8); o = np.zeros(12, np.uint64)
        rc = self.lib.bno_g1_decode(b.ctypes.data_as(C.c_void_p), _p(o)); return rc, o
    def g2_decode(self, b):
        b = np.ascontiguousarray(b, np.uint8); o = np.zeros(24, np.uint64)
        rc = self.lib.bno_g2_decode(b.ctypes.data_as(C.c_void_p), _p(o)); return rc, o

    # ---- conveniences
    def fq12_from_ints(self, v): return np.concatenate([self.fp_from_int(FQ, int(x)) for x in v])
    def fq12_to_ints(self, a): return [self.fp_to_int(FQ, a[4 * i:4 * i + 4]) for i in range(12)]
    def fq2_from_ints(self, v): return np.concatenate([self.fp_from_int(FQ, int(x)) for x in v])

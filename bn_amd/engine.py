"""Batch engine over the C ABI.  Arrays are numpy uint64 in the reference's #[repr(C)] layouts:
   Fr (n,4)  G1 (n,12)  G2 (n,24)  Gt (n,48)   - Montgomery limbs, little endian."""
import ctypes as C

import numpy as np

from . import _native

FR_BYTES, G1_WORDS, G2_WORDS, GT_WORDS = 32, 12, 24, 48


def _arr(a, width):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.ndim != 2 or a.shape[1] != width:
        raise ValueError(f"expected shape (n,{width}) uint64, got {a.shape}")
    return a


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _same_len(a, b):
    if a.shape[0] != b.shape[0]:
        raise ValueError(f"operands differ in length: {a.shape[0]} vs {b.shape[0]}")


class Engine:
    """one context = one GPU (include/bn254_hip.h: bn254_ctx)"""

    def __init__(self, device=0):
        self._lib = _native.lib()
        if self._lib.bn254_device_count() <= 0:
            raise _native.Bn254Error("no HIP device: bn_amd has no CPU fallback")
        h = C.c_void_p()
        _native.check(self._lib.bn254_ctx_create(int(device), C.byref(h)))
        self._ctx = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.bn254_ctx_destroy(self._ctx)
            self._ctx = None

    # ---- tunables (include/bn254_hip.h BN254_OPT_*): names as in _native.OPTIONS; None / a negative value restores the default
    def set_option(self, name, value):
        _native.check(self._lib.bn254_ctx_set_option(self._h, _native.OPTIONS[name], -1 if value is None else int(value)))

    def get_option(self, name):
        v = C.c_long()
        _native.check(self._lib.bn254_ctx_get_option(self._h, _native.OPTIONS[name], C.byref(v)))
        return v.value

    def get_option_raw(self, name):
        """the explicitly set value of an option, or None while its default is in effect"""
        v = C.c_long()
        _native.check(self._lib.bn254_ctx_get_option_raw(self._h, _native.OPTIONS[name], C.byref(v)))
        return None if v.value < 0 else v.value

    def options(self, **kw):
        """context manager: set the given options, restore the previous RAW state (default or explicit) on exit"""
        eng = self

        class _Scope:
            def _restore(self_):
                for k, raw in self_.saved.items():
                    eng.set_option(k, raw)

            def __enter__(self_):
                self_.saved = {k: eng.get_option_raw(k) for k in kw}          # name -> None (default) or the explicit value: read, never written
                try:
                    for k, v in kw.items():
                        eng.set_option(k, v)
                except Exception:                  # a rejected value (BN254_E_BAD_ARG) must not leave the context half-configured
                    self_._restore()
                    raise
                return eng

            def __exit__(self_, *exc):
                self_._restore()
                return False
        return _Scope()

    @property
    def _h(self):
        """the context handle; a closed engine raises instead of silently falling back to the C ABI's NULL = default context"""
        if self._ctx is None:
            raise _native.Bn254Error("this Engine is closed")
        return self._ctx

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer API
    def pairing_batch(self, p, q, out=None):
        """out: optional preallocated (n,48) uint64 array (a long-running caller reuses it; a fresh np.empty pays one page fault
        per 4 KB inside the D2H copy)"""
        p = _arr(p, G1_WORDS); q = _arr(q, G2_WORDS)
        if p.shape[0] != q.shape[0]:
            raise ValueError("p and q differ in length")
        if out is None:
            out = np.empty((p.shape[0], GT_WORDS), np.uint64)
        elif out.shape != (p.shape[0], GT_WORDS) or out.dtype != np.uint64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (n,48) uint64 array")
        _native.check(self._lib.bn254_pairing_batch(self._h, _p(p), _p(q), _p(out), p.shape[0]))
        return out

    def pairing_product(self, p, q):
        p = _arr(p, G1_WORDS) if len(p) else np.zeros((0, G1_WORDS), np.uint64)
        q = _arr(q, G2_WORDS) if len(q) else np.zeros((0, G2_WORDS), np.uint64)
        if p.shape[0] != q.shape[0]:
            raise ValueError("p and q differ in length")
        out = np.empty(GT_WORDS, np.uint64)
        _native.check(self._lib.bn254_pairing_product(self._h, _p(p), _p(q), p.shape[0], _p(out)))
        return out

    def g1_mul_batch(self, p, k):
        p = _arr(p, G1_WORDS); k = _arr(k, 4); _same_len(p, k)
        out = np.empty_like(p)
        _native.check(self._lib.bn254_g1_mul_batch(self._h, _p(p), _p(k), _p(out), p.shape[0]))
        return out

    def g2_mul_batch(self, p, k):
        p = _arr(p, G2_WORDS); k = _arr(k, 4); _same_len(p, k)
        out = np.empty_like(p)
        _native.check(self._lib.bn254_g2_mul_batch(self._h, _p(p), _p(k), _p(out), p.shape[0]))
        return out

    def g1_add_batch(self, a, b, negate_b=False):
        """a + b (a - b): the reference's Jacobian limbs (groups/mod.rs:275-347)"""
        a = _arr(a, G1_WORDS); b = _arr(b, G1_WORDS); _same_len(a, b); out = np.empty_like(a)
        _native.check(self._lib.bn254_g1_add_batch(self._h, _p(a), _p(b), _p(out), a.shape[0], 1 if negate_b else 0)); return out

    def g2_add_batch(self, a, b, negate_b=False):
        a = _arr(a, G2_WORDS); b = _arr(b, G2_WORDS); _same_len(a, b); out = np.empty_like(a)
        _native.check(self._lib.bn254_g2_add_batch(self._h, _p(a), _p(b), _p(out), a.shape[0], 1 if negate_b else 0)); return out

    def g2_precompute(self, q):
        """(n,24) G2 -> (n,102,24) line coefficients [ell_0 | ell_vw | ell_vv] (groups/mod.rs:557-588)"""
        q = _arr(q, G2_WORDS)
        out = np.empty((q.shape[0], 102, 24), np.uint64)
        _native.check(self._lib.bn254_g2_precompute(self._h, _p(q), _p(out), q.shape[0]))
        return out

    def pairing_prepared_batch(self, p, coeffs):
        """coeffs (102,24): shared by all p;  (n,102,24): one prepared point per p"""
        p = _arr(p, G1_WORDS)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        shared = coeffs.ndim == 2
        if coeffs.shape[-2:] != (102, 24) or (not shared and coeffs.shape[0] != p.shape[0]):
            raise ValueError("coeffs must be (102,24) or (n,102,24)")
        out = np.empty((p.shape[0], GT_WORDS), np.uint64)
        _native.check(self._lib.bn254_pairing_prepared_batch(self._h, _p(p), _p(coeffs), 1 if shared else 0, _p(out), p.shape[0]))
        return out

    # ---- native prepared-G2 mode (include/bn254_hip.h bn254_g2_prepare ...)
    def g2_prepare(self, q):
        """(n,24) G2 points (or one (24,) point) -> PreparedG2: the device-resident native line tables (one point: shared by every P)"""
        q = _arr(q, G2_WORDS)
        h = C.c_void_p()
        _native.check(self._lib.bn254_g2_prepare(self._h, _p(q), q.shape[0], C.byref(h)))
        return PreparedG2(self, h)

    def g2_prepare_dev(self, d_q, n, stream=0):
        h = C.c_void_p()
        _native.check(self._lib.bn254_g2_prepare_dev(self._h, d_q, n, C.byref(h), stream))
        return PreparedG2(self, h)

    def pairing_prepared_native_batch(self, p, prepared, out=None):
        """out[i] = pairing(p[i], Q) for a one-point handle, pairing(p[i], Q[i]) otherwise"""
        p = _arr(p, G1_WORDS)
        if out is None:
            out = np.empty((p.shape[0], GT_WORDS), np.uint64)
        _native.check(self._lib.bn254_pairing_prepared_native_batch(self._h, _p(p), prepared._h, _p(out), p.shape[0]))
        return out

    def pairing_product_prepared_native(self, p, prepared):
        """fold(Gt::one(), acc * pairing(p[i], Q[i])) over prepared points (one-point handle: against that point) -> (48,) uint64"""
        p = _arr(p, G1_WORDS) if len(p) else np.zeros((0, G1_WORDS), np.uint64)
        out = np.empty(GT_WORDS, np.uint64)
        _native.check(self._lib.bn254_pairing_product_prepared_native(self._h, _p(p), prepared._h, p.shape[0], _p(out)))
        return out

    def miller_product_prepared_native_dev(self, d_p, prepared, n, d_partial, q_first=0, stream=0):
        _native.check(self._lib.bn254_miller_product_prepared_native_dev(self._h, d_p, prepared._h, q_first, n, d_partial, stream))

    def miller_prepared_native_dev(self, d_p, prepared, d_f, n, q_first=0, stream=0):
        _native.check(self._lib.bn254_miller_prepared_native_dev(self._h, d_p, prepared._h, q_first, d_f, n, stream))

    def pairing_prepared_native_dev(self, d_p, prepared, d_out, n, q_first=0, stream=0):
        _native.check(self._lib.bn254_pairing_prepared_native_batch_dev(self._h, d_p, prepared._h, q_first, d_out, n, stream))

    # ---- wire format (fixed-size records: G1 65 bytes, G2 129 bytes)
    def fr_encode_batch(self, k):
        k = _arr(k, 4); out = np.empty((k.shape[0], 32), np.uint8)
        _native.check(self._lib.bn254_fr_encode_batch(self._h, _p(k), _p(out), k.shape[0])); return out

    def fr_decode_batch(self, b):
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32); n = b.shape[0]
        out = np.empty((n, 4), np.uint64); st = np.empty(n, np.int32)
        _native.check(self._lib.bn254_fr_decode_batch(self._h, _p(b), _p(out), _p(st), n)); return out, st

    def g1_encode_batch(self, p):
        p = _arr(p, G1_WORDS); out = np.empty((p.shape[0], 65), np.uint8)
        _native.check(self._lib.bn254_g1_encode_batch(self._h, _p(p), _p(out), p.shape[0])); return out

    def g2_encode_batch(self, p):
        p = _arr(p, G2_WORDS); out = np.empty((p.shape[0], 129), np.uint8)
        _native.check(self._lib.bn254_g2_encode_batch(self._h, _p(p), _p(out), p.shape[0])); return out

    def g1_decode_batch(self, b):
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 65); n = b.shape[0]
        out = np.empty((n, G1_WORDS), np.uint64); st = np.empty(n, np.int32)
        _native.check(self._lib.bn254_g1_decode_batch(self._h, _p(b), _p(out), _p(st), n)); return out, st

    def g2_decode_batch(self, b):
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 129); n = b.shape[0]
        out = np.empty((n, G2_WORDS), np.uint64); st = np.empty(n, np.int32)
        _native.check(self._lib.bn254_g2_decode_batch(self._h, _p(b), _p(out), _p(st), n)); return out, st

    # ---- the crate's variable-length byte stream (infinity = the lone byte 0)
    def _encode_stream(self, fn, p, rec):
        out = np.empty(p.shape[0] * rec, np.uint8); w = C.c_size_t()
        _native.check(fn(self._h, _p(p), p.shape[0], _p(out), out.size, C.byref(w)))
        return out[:w.value].copy()

    def g1_encode_stream(self, p):
        return self._encode_stream(self._lib.bn254_g1_encode_stream, _arr(p, G1_WORDS), 65)

    def g2_encode_stream(self, p):
        return self._encode_stream(self._lib.bn254_g2_encode_stream, _arr(p, G2_WORDS), 129)

    def _decode_stream(self, fn, b, words, max_points):
        b = np.ascontiguousarray(b, np.uint8).reshape(-1)
        cap = b.size if max_points is None else int(max_points)
        out = np.zeros((max(cap, 1), words), np.uint64); st = np.zeros(max(cap, 1), np.int32); cnt = C.c_size_t(); used = C.c_size_t()
        _native.check(fn(self._h, _p(b), b.size, _p(out), _p(st), cap, C.byref(cnt), C.byref(used)))
        return out[:cnt.value], st[:cnt.value], used.value

    def g1_decode_stream(self, b, max_points=None):
        """bytes -> (points, status, bytes consumed)"""
        return self._decode_stream(self._lib.bn254_g1_decode_stream, b, G1_WORDS, max_points)

    def g2_decode_stream(self, b, max_points=None):
        return self._decode_stream(self._lib.bn254_g2_decode_stream, b, G2_WORDS, max_points)

    def gt_mul_batch(self, a, b):
        a = _arr(a, GT_WORDS); b = _arr(b, GT_WORDS); _same_len(a, b)
        out = np.empty_like(a)
        _native.check(self._lib.bn254_gt_mul_batch(self._h, _p(a), _p(b), _p(out), a.shape[0]))
        return out

    def gt_pow_batch(self, a, k):
        a = _arr(a, GT_WORDS); k = _arr(k, 4); _same_len(a, k)
        out = np.empty_like(a)
        _native.check(self._lib.bn254_gt_pow_batch(self._h, _p(a), _p(k), _p(out), a.shape[0]))
        return out

    def gt_inverse_batch(self, a):
        """Gt::inverse (lib.rs:172)"""
        a = _arr(a, GT_WORDS)
        out = np.empty_like(a)
        _native.check(self._lib.bn254_gt_inverse_batch(self._h, _p(a), _p(out), a.shape[0]))
        return out

    # ---- device-resident API: raw device pointers (ints) + hipStream_t (int or 0)
    def pairing_batch_dev(self, d_p, d_q, d_out, n, stream=0):
        _native.check(self._lib.bn254_pairing_batch_dev(self._h, d_p, d_q, d_out, n, stream))

    def miller_batch_dev(self, d_p, d_q, d_f, n, stream=0):
        _native.check(self._lib.bn254_miller_batch_dev(self._h, d_p, d_q, d_f, n, stream))

    def final_exp_batch_dev(self, d_f, d_out, n, stream=0):
        _native.check(self._lib.bn254_final_exp_batch_dev(self._h, d_f, d_out, n, stream))

    def gt_product_dev(self, d_in, n, d_out, stream=0):
        _native.check(self._lib.bn254_gt_product_dev(self._h, d_in, n, d_out, stream))

    def gt_product_final_exp_dev(self, d_in, m, d_out, stream=0):
        _native.check(self._lib.bn254_gt_product_final_exp_dev(self._h, d_in, m, d_out, stream))

    def miller_product_dev(self, d_p, d_q, n, d_partial, stream=0):
        _native.check(self._lib.bn254_miller_product_dev(self._h, d_p, d_q, n, d_partial, stream))

    def g2_precompute_dev(self, d_q, d_coeffs, n, stream=0):
        _native.check(self._lib.bn254_g2_precompute_dev(self._h, d_q, d_coeffs, n, stream))

    def miller_prepared_dev(self, d_p, d_coeffs, shared, d_f, n, stream=0):
        _native.check(self._lib.bn254_miller_prepared_dev(self._h, d_p, d_coeffs, 1 if shared else 0, d_f, n, stream))

    def g1_mul_dev(self, d_p, d_k, d_out, n, stream=0, normalize=True):
        f = self._lib.bn254_g1_mul_batch_dev if normalize else self._lib.bn254_g1_mul_jacobian_dev
        _native.check(f(self._h, d_p, d_k, d_out, n, stream))

    def g2_mul_dev(self, d_p, d_k, d_out, n, stream=0, normalize=True):
        f = self._lib.bn254_g2_mul_batch_dev if normalize else self._lib.bn254_g2_mul_jacobian_dev
        _native.check(f(self._h, d_p, d_k, d_out, n, stream))

    def gt_mul_dev(self, d_a, d_b, d_out, n, stream=0):
        _native.check(self._lib.bn254_gt_mul_batch_dev(self._h, d_a, d_b, d_out, n, stream))

    def gt_pow_dev(self, d_a, d_k, d_out, n, stream=0):
        _native.check(self._lib.bn254_gt_pow_batch_dev(self._h, d_a, d_k, d_out, n, stream))

    def exp_by_neg_z_dev(self, d_in, d_out, n, stream=0):
        """Fq12::exp_by_neg_z as the reference writes it (fq12.rs:229-246), any Fq12 in"""
        _native.check(self._lib.bn254_exp_by_neg_z_dev(self._h, d_in, d_out, n, stream))

    def synthetic_scalars_dev(self, seed, lo, n, which, d_out, stream=0):
        _native.check(self._lib.bn254_synthetic_scalars_dev(self._h, seed, lo, n, which, d_out, stream))

    def tile_dev(self, d_record, record_bytes, n, d_out, stream=0):
        _native.check(self._lib.bn254_tile_dev(self._h, d_record, record_bytes, n, d_out, stream))

    # ---- measurement
    def ubench_mac32(self, waves_per_simd=8, iters=1 << 15, operand_bits=32):
        """(G lane-MAC32 per second of a pure v_mad_u64_u32 stream, kernel ms) - the same-run `roofline.peak` of bench.py;
        operand_bits = 29: on the engine's own limbs (the multiplier's rate depends on its data)"""
        g = C.c_double(); ms = C.c_double()
        _native.check(self._lib.bn254_ubench_mac32_ex(self._h, int(waves_per_simd), int(iters), int(operand_bits), C.byref(g), C.byref(ms)))
        return g.value, ms.value

    def wave_ubench(self, which, iters=200):
        """microseconds per run of one program of the wave-cooperative machine on a single wave"""
        ms = C.c_double()
        _native.check(self._lib.bn254_wave_ubench(self._h, int(which), int(iters), C.byref(ms)))
        return ms.value * 1e3 / iters

    def profile(self, on=True):
        _native.check(self._lib.bn254_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        _native.check(self._lib.bn254_profile_reset(self._h))

    def kernel_stats(self, kernel):
        ms = C.c_double(); cnt = C.c_uint64()
        _native.check(self._lib.bn254_kernel_stats(self._h, kernel.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


class PreparedG2:
    """opaque handle of bn254_g2_prepare: the native line tables of `count` G2 points in device memory (33 792 B each)"""

    def __init__(self, engine, handle):
        self._eng = engine
        self._p = handle

    @property
    def _h(self):
        if self._p is None:
            raise _native.Bn254Error("this PreparedG2 is closed")
        return self._p

    @property
    def count(self):
        return int(self._eng._lib.bn254_g2_prepared_count(self._h))

    @property
    def device_bytes(self):
        return int(self._eng._lib.bn254_g2_prepared_bytes(self._h))

    def export(self):
        """the table as (88 lines, 12 groups, 2 x count columns, 4) uint32 - for tests"""
        n = self.count
        out = np.empty((88, 12, 2 * n, 4), np.uint32)
        _native.check(self._eng._lib.bn254_g2_prepared_export(self._eng._h, self._h, _p(out), out.nbytes))
        return out

    def close(self):
        if getattr(self, "_p", None) and getattr(self._eng, "_ctx", None):
            self._eng._lib.bn254_g2_prepared_destroy(self._p)
        self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiEngine:
    """several GPUs of one node behind ONE host process (include/bn254_hip.h: bn254_multi): contiguous shards of independent
    pairings, and the multi-pairing product with its single 384-byte-per-rank exchange (RCCL all-gather when every rank has
    its own GPU, peer copies when a device is listed twice)."""

    def __init__(self, devices, exchange="auto"):
        """exchange: "auto" (RCCL when every rank has its own GPU and RCCL loads, else peer copies), "peer", "rccl" (fail instead of
        falling back)"""
        self._lib = _native.lib()
        if self._lib.bn254_device_count() <= 0:
            raise _native.Bn254Error("no HIP device: bn_amd has no CPU fallback")
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        _native.check(self._lib.bn254_multi_create_ex(devs, len(devices), _native.EXCHANGE[exchange], C.byref(h)))
        self._m = h
        self.devices = list(devices)

    def set_option(self, name, value):
        _native.check(self._lib.bn254_multi_set_option(self._h, _native.OPTIONS[name], -1 if value is None else int(value)))

    @property
    def _h(self):
        if self._m is None:
            raise _native.Bn254Error("this MultiEngine is closed")
        return self._m

    @property
    def exchange(self):
        return {0: "peer", 1: "rccl"}[self._lib.bn254_multi_exchange_kind(self._h)]

    @property
    def numa_nodes(self):
        """per rank: the NUMA node its host thread is pinned to during a call, -1 = not pinned"""
        return [self._lib.bn254_multi_rank_numa_node(self._h, g) for g in range(len(self.devices))]

    def close(self):
        if getattr(self, "_m", None):
            self._lib.bn254_multi_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pairing_batch(self, p, q, out=None):
        p = _arr(p, G1_WORDS); q = _arr(q, G2_WORDS); _same_len(p, q)
        if out is None:
            out = np.empty((p.shape[0], GT_WORDS), np.uint64)
        elif out.shape != (p.shape[0], GT_WORDS) or out.dtype != np.uint64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (n,48) uint64 array")
        _native.check(self._lib.bn254_pairing_batch_multi(self._h, _p(p), _p(q), _p(out), p.shape[0]))
        return out

    def pairing_product(self, p, q):
        p = _arr(p, G1_WORDS) if len(p) else np.zeros((0, G1_WORDS), np.uint64)
        q = _arr(q, G2_WORDS) if len(q) else np.zeros((0, G2_WORDS), np.uint64)
        _same_len(p, q)
        out = np.empty(GT_WORDS, np.uint64)
        _native.check(self._lib.bn254_pairing_product_multi(self._h, _p(p), _p(q), p.shape[0], _p(out)))
        return out

    def g2_prepare(self, q):
        """one point: prepared on every rank's GPU; n points: sharded over the ranks (then paired with exactly n points p)"""
        q = _arr(q, G2_WORDS)
        h = C.c_void_p()
        _native.check(self._lib.bn254_g2_prepare_multi(self._h, _p(q), q.shape[0], C.byref(h)))
        return MultiPreparedG2(self, h)

    def pairing_prepared_native_batch(self, p, prepared, out=None):
        p = _arr(p, G1_WORDS)
        if out is None:
            out = np.empty((p.shape[0], GT_WORDS), np.uint64)
        _native.check(self._lib.bn254_pairing_prepared_native_batch_multi(self._h, _p(p), prepared._h, _p(out), p.shape[0]))
        return out

    def pairing_product_prepared_native(self, p, prepared):
        """the multi-pairing over the prepared points, sharded: one 384-byte exchange, ONE final exponentiation"""
        p = _arr(p, G1_WORDS) if len(p) else np.zeros((0, G1_WORDS), np.uint64)
        out = np.empty(GT_WORDS, np.uint64)
        _native.check(self._lib.bn254_pairing_product_prepared_native_multi(self._h, _p(p), prepared._h, p.shape[0], _p(out)))
        return out


class MultiPreparedG2:
    """handle of bn254_g2_prepare_multi: per-rank native tables"""

    def __init__(self, multi, handle):
        self._m = multi
        self._p = handle

    @property
    def _h(self):
        if self._p is None:
            raise _native.Bn254Error("this MultiPreparedG2 is closed")
        return self._p

    @property
    def count(self):
        return int(self._m._lib.bn254_multi_prepared_count(self._h))

    def close(self):
        if getattr(self, "_p", None) and getattr(self._m, "_m", None):
            self._m._lib.bn254_multi_prepared_destroy(self._p)
        self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

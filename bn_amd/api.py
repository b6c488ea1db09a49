"""Host-side mirror of the reference crate's public API (src/lib.rs): Fr, G1, G2, Gt, pairing - plus the batch entry points
the GPU engine adds (pairing_batch / pairing_product; the reference only has the fold of shootout/main.rs:11-16).

Values are immutable wrappers around the reference's memory images (Montgomery u64 limbs).  Group and pairing arithmetic
runs on the GPU through the C ABI; only the scalar field Fr (lib.rs:15-53, "host-side convenience" in SURVEY.md) is plain
Python integer arithmetic.  No CPU fallback for anything that touches a curve point or a Gt."""
import numpy as np

from .engine import Engine, G1_WORDS, G2_WORDS, GT_WORDS

_U = 4965661367192848881
Q_MOD = 36 * _U**4 + 36 * _U**3 + 24 * _U**2 + 6 * _U + 1
R_MOD = 36 * _U**4 + 36 * _U**3 + 18 * _U**2 + 6 * _U + 1
_MONT = 1 << 256
_M64 = (1 << 64) - 1

_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


def _limbs(v):
    return np.array([(v >> (64 * i)) & _M64 for i in range(4)], np.uint64)


def _mont(v, mod):
    return _limbs(v % mod * _MONT % mod)


class Fr:
    """scalar field element (lib.rs:15-53); `limbs` is the reference's Montgomery image"""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = int(v) % R_MOD

    @staticmethod
    def zero(): return Fr(0)
    @staticmethod
    def one(): return Fr(1)
    @staticmethod
    def from_str(s):                      # lib.rs:24 -> fields/fp.rs:39-59: ASCII decimal digits only; "" is zero; anything else None
        if not all(c in "0123456789" for c in s):
            return None
        return Fr(int(s)) if s else Fr(0)
    @staticmethod
    def random(rng):                      # uniform mod r from 512 bits, like arith.rs:195-198
        return Fr(int.from_bytes(rng.bytes(64), "little"))
    @staticmethod
    def from_limbs(l):
        return Fr(sum(int(x) << (64 * i) for i, x in enumerate(l)) * pow(_MONT, -1, R_MOD))
    @property
    def limbs(self): return _mont(self.v, R_MOD)
    def inverse(self): return None if self.v == 0 else Fr(pow(self.v, -1, R_MOD))
    def is_zero(self): return self.v == 0
    def pow(self, e): return Fr(pow(self.v, e.v, R_MOD))
    def __add__(self, o): return Fr(self.v + o.v)
    def __sub__(self, o): return Fr(self.v - o.v)
    def __mul__(self, o): return Fr(self.v * o.v)
    def __neg__(self): return Fr(-self.v)
    def __eq__(self, o): return isinstance(o, Fr) and self.v == o.v
    def __hash__(self): return hash(self.v)
    def __repr__(self): return f"Fr({self.v})"


def _one_fq(): return _mont(1, Q_MOD)


class _Point:
    WORDS = 0
    __slots__ = ("limbs",)

    def __init__(self, limbs):
        self.limbs = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(self.WORDS)

    def is_zero(self):                    # groups/mod.rs:224-226: z == 0
        return not self.limbs[2 * self.WORDS // 3:].any()

    def __eq__(self, o):                  # projective equality (groups/mod.rs:83-109) via the normalized images
        return type(o) is type(self) and np.array_equal(self.normalize().limbs, o.normalize().limbs)

    def normalize(self):                  # lib.rs:88-95 / 131-138: multiply by one on the GPU and normalize there
        return self * Fr.one()


class G1(_Point):
    WORDS = G1_WORDS

    @staticmethod
    def one():                            # groups/mod.rs:355-361: (1, 2, 1)
        return G1(np.concatenate([_one_fq(), _mont(2, Q_MOD), _one_fq()]))
    @staticmethod
    def zero():                           # groups/mod.rs:208-214: (0, 1, 0)
        return G1(np.concatenate([np.zeros(4, np.uint64), _one_fq(), np.zeros(4, np.uint64)]))
    @staticmethod
    def random(rng): return G1.one() * Fr.random(rng)        # groups/mod.rs:220-222
    def __mul__(self, k):                 # lib.rs:116-120 (result returned normalized)
        return G1(default_engine().g1_mul_batch(self.limbs, k.limbs)[0])
    def __add__(self, o):                 # lib.rs:103-106 (the reference's Jacobian limbs)
        return G1(default_engine().g1_add_batch(self.limbs, o.limbs)[0])
    def __sub__(self, o):                 # lib.rs:108-111
        return G1(default_engine().g1_add_batch(self.limbs, o.limbs, negate_b=True)[0])
    def __neg__(self):                    # lib.rs:113-114
        return G1(default_engine().g1_add_batch(G1.zero().limbs, self.limbs, negate_b=True)[0])


_G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634),
           (8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531))


class G2(_Point):
    WORDS = G2_WORDS

    @staticmethod
    def one():                            # groups/mod.rs:377-390
        (x0, x1), (y0, y1) = _G2_GEN
        return G2(np.concatenate([_mont(x0, Q_MOD), _mont(x1, Q_MOD), _mont(y0, Q_MOD), _mont(y1, Q_MOD), _one_fq(), np.zeros(4, np.uint64)]))
    @staticmethod
    def zero():
        z = np.zeros(8, np.uint64)
        return G2(np.concatenate([z, _one_fq(), np.zeros(4, np.uint64), z]))
    @staticmethod
    def random(rng): return G2.one() * Fr.random(rng)
    def __mul__(self, k):
        return G2(default_engine().g2_mul_batch(self.limbs, k.limbs)[0])
    def __add__(self, o):                 # lib.rs:146-149
        return G2(default_engine().g2_add_batch(self.limbs, o.limbs)[0])
    def __sub__(self, o):
        return G2(default_engine().g2_add_batch(self.limbs, o.limbs, negate_b=True)[0])
    def __neg__(self):
        return G2(default_engine().g2_add_batch(G2.zero().limbs, self.limbs, negate_b=True)[0])


class Gt:
    """target group element (lib.rs:165-179)"""
    __slots__ = ("limbs",)

    def __init__(self, limbs):
        self.limbs = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(GT_WORDS)

    @staticmethod
    def one():
        l = np.zeros(GT_WORDS, np.uint64); l[:4] = _one_fq()
        return Gt(l)
    def __mul__(self, o):                 # lib.rs:175-179
        return Gt(default_engine().gt_mul_batch(self.limbs, o.limbs)[0])
    def pow(self, k):                     # lib.rs:171
        return Gt(default_engine().gt_pow_batch(self.limbs, k.limbs)[0])
    def inverse(self):                    # lib.rs:172
        return Gt(default_engine().gt_inverse_batch(self.limbs)[0])
    def __eq__(self, o): return isinstance(o, Gt) and np.array_equal(self.limbs, o.limbs)   # canonical limbs: memcmp
    def __repr__(self): return "Gt(%s...)" % hex(int(self.limbs[0]))


def pairing(p, q):
    """lib.rs:181-183"""
    return Gt(default_engine().pairing_batch(p.limbs, q.limbs)[0])


def pairing_batch(ps, qs, engine=None):
    """out[i] = pairing(ps[i], qs[i]); ps/qs: sequences of G1/G2 or (n,12)/(n,24) uint64 arrays"""
    e = engine or default_engine()
    P = np.stack([p.limbs for p in ps]) if not isinstance(ps, np.ndarray) else ps
    Q = np.stack([q.limbs for q in qs]) if not isinstance(qs, np.ndarray) else qs
    return e.pairing_batch(P, Q)


def pairing_product(ps, qs, engine=None):
    """fold(Gt::one(), acc * pairing(p, q)) (shootout/main.rs:11-16) with ONE final exponentiation"""
    e = engine or default_engine()
    P = np.stack([p.limbs for p in ps]) if not isinstance(ps, np.ndarray) else ps
    Q = np.stack([q.limbs for q in qs]) if not isinstance(qs, np.ndarray) else qs
    return Gt(e.pairing_product(P, Q))


class PreparedG2:
    """G2 points prepared once for many pairings (the crate's internal G2Precomp, groups/mod.rs:472-483,557-588, as a device-resident
    native table: Engine.g2_prepare).  One point: shared by every P; several: point i is paired with ps[i]."""

    def __init__(self, qs, engine=None):
        e = engine or default_engine()
        Q = qs.limbs if isinstance(qs, G2) else (np.stack([q.limbs for q in qs]) if not isinstance(qs, np.ndarray) else qs)
        self._e = e
        self._h = e.g2_prepare(Q)

    def __len__(self):
        return self._h.count

    def pairing(self, p):
        """== pairing(p, q) for a one-point handle"""
        return Gt(self._e.pairing_prepared_native_batch(p.limbs, self._h)[0])

    def pairing_batch(self, ps):
        P = np.stack([p.limbs for p in ps]) if not isinstance(ps, np.ndarray) else ps
        return self._e.pairing_prepared_native_batch(P, self._h)

    def pairing_product(self, ps):
        """== fold(Gt::one(), acc * pairing(ps[i], q[i])) (shootout/main.rs:11-16) with ONE final exponentiation"""
        P = np.stack([p.limbs for p in ps]) if not isinstance(ps, np.ndarray) else ps
        return Gt(self._e.pairing_product_prepared_native(P, self._h))

    def close(self):
        self._h.close()

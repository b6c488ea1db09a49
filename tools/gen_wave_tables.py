#!/usr/bin/env python3
"""Generator of the WAVE-COOPERATIVE Fq12 machine (bn_amd/csrc/wave.hpp): role tables and programs -> bn_amd/csrc/wave_tables.hpp.

Why it exists: a single final exponentiation on one lane pair is a 2.1 ms serial chain, and the last
levels of the multi-pairing product tree are the same shape.  Here ONE Fq12 operation is spread over the 32 lane pairs of a wave:
every Fq2 product of an Fq12 product (18, Karatsuba) or of a Granger-Scott squaring (9 Fq2 squarings) runs on its own lane pair
at the same time, the operands are gathered from - and the results recombined through - a register file of Fq2 values in LDS.

The machine has two kinds of phases, both table driven (one 24-byte role per lane pair and phase, so the code is lane-uniform):
    PROD   R[td] = (sum of up to 4 registers, optionally conjugated) * (sum of up to 4 registers)      or the square of the first
    COMB   R[dst] = reduce( CX * xi * (R[x0] - R[x1] - R[x2])  +  CY * (R[y0] + R[y1] - R[y2] - R[y3])  +  zs * CZ * R[z] )
                    (CX, CY, CZ) = (1, 1, 0) "M" for products, (3, 3, 2) "C" for the cyclotomic squaring
plus INV (Fq2 inversion of a gathered value on every pair).  A program is a list of (op, phase table, base register); register
indices flagged REL are relative to the base (table slots of the exponentiation, operands of a product).

This file is a BUILD tool: it executes every program it emits on exact field elements (its own small big-integer model of the
tower, fq2.rs / fq6.rs / fq12.rs formulas) and asserts the result - Fq12 product, cyclotomic squaring, Frobenius maps, the whole
final exponentiation of fq12.rs:41-88 - before writing the header; it also checks that no phase reads a register another pair
writes in the same phase.  tests/test_wave_tables.py repeats the value checks against oracle/bn_model.py.
"""
import pathlib
import random
import sys

U = 4965661367192848881
Q = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
R_ORD = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1
XI = (9, 1)

# ------------------------------------------------------------------------------------------------ exact model (fields/*.rs)
def f2_add(x, y): return ((x[0] + y[0]) % Q, (x[1] + y[1]) % Q)
def f2_sub(x, y): return ((x[0] - y[0]) % Q, (x[1] - y[1]) % Q)
def f2_neg(x): return ((-x[0]) % Q, (-x[1]) % Q)
def f2_mul(x, y): return ((x[0] * y[0] - x[1] * y[1]) % Q, (x[0] * y[1] + x[1] * y[0]) % Q)
def f2_scale(x, k): return (x[0] * k % Q, x[1] * k % Q)
def f2_conj(x): return (x[0], (-x[1]) % Q)
def f2_inv(x):
    n = pow((x[0] * x[0] + x[1] * x[1]) % Q, Q - 2, Q)
    return (x[0] * n % Q, (-x[1]) * n % Q)
def f2_pow(x, e):
    r = (1, 0)
    while e:
        if e & 1: r = f2_mul(r, x)
        x = f2_mul(x, x); e >>= 1
    return r
F2_ZERO, F2_ONE = (0, 0), (1, 0)
def gamma(num, den, p): return f2_pow(XI, (Q**p - 1) * num // den)
FROB6_C1 = [gamma(1, 3, p) for p in range(4)]
FROB6_C2 = [gamma(2, 3, p) for p in range(4)]
FROB12_C1 = [gamma(1, 6, p) for p in range(4)]
# an Fq12 as a flat list of six Fq2: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 (the order of the 384-byte image)
def f6_mul(a, b):
    t = [F2_ZERO] * 5
    for i in range(3):
        for j in range(3):
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    return [f2_add(t[0], f2_mul(XI, t[3])), f2_add(t[1], f2_mul(XI, t[4])), t[2]]
def f6_add(a, b): return [f2_add(x, y) for x, y in zip(a, b)]
def f6_sub(a, b): return [f2_sub(x, y) for x, y in zip(a, b)]
def f6_mul_v(a): return [f2_mul(XI, a[2]), a[0], a[1]]
def f6_inv(a):
    c0 = f2_sub(f2_mul(a[0], a[0]), f2_mul(XI, f2_mul(a[1], a[2])))
    c1 = f2_sub(f2_mul(XI, f2_mul(a[2], a[2])), f2_mul(a[0], a[1]))
    c2 = f2_sub(f2_mul(a[1], a[1]), f2_mul(a[0], a[2]))
    n = f2_add(f2_mul(XI, f2_add(f2_mul(a[2], c1), f2_mul(a[1], c2))), f2_mul(a[0], c0))
    t = f2_inv(n)
    return [f2_mul(t, c0), f2_mul(t, c1), f2_mul(t, c2)]
def f12_mul(a, b):
    aa, bb = f6_mul(a[:3], b[:3]), f6_mul(a[3:], b[3:])
    t = f6_mul(f6_add(a[:3], a[3:]), f6_add(b[:3], b[3:]))
    return f6_add(aa, f6_mul_v(bb)) + f6_sub(f6_sub(t, aa), bb)
def f12_conj(a): return a[:3] + [f2_neg(x) for x in a[3:]]
def f12_inv(a):
    d = f6_sub(f6_mul(a[:3], a[:3]), f6_mul_v(f6_mul(a[3:], a[3:])))
    t = f6_inv(d)
    return f6_mul(a[:3], t) + [f2_neg(x) for x in f6_mul(a[3:], t)]
def f6_frob(a, p):
    c = (lambda x: f2_conj(x)) if p & 1 else (lambda x: x)
    return [c(a[0]), f2_mul(c(a[1]), FROB6_C1[p]), f2_mul(c(a[2]), FROB6_C2[p])]
def f12_frob(a, p): return f6_frob(a[:3], p) + [f2_mul(x, FROB12_C1[p]) for x in f6_frob(a[3:], p)]
def f12_pow(a, e):
    r = [F2_ONE] + [F2_ZERO] * 5
    while e:
        if e & 1: r = f12_mul(r, a)
        a = f12_mul(a, a); e >>= 1
    return r
def f12_one(): return [F2_ONE] + [F2_ZERO] * 5
def final_exponentiation(f):                                   # fq12.rs:41-88, written with plain powers (the value is what counts)
    b = f12_inv(f)
    c = f12_mul(f12_conj(f), b)
    s = f12_mul(f12_frob(c, 2), c)                             # first chunk
    ez = lambda x: f12_conj(f12_pow(x, U))                     # exp_by_neg_z on the cyclotomic subgroup
    cs = lambda x: f12_mul(x, x)
    a = ez(s); b = cs(a); c = cs(b); d = f12_mul(c, b); e = ez(d); ff = cs(e); g = ez(ff); h = f12_conj(d); i = f12_conj(g)
    j = f12_mul(i, e); k = f12_mul(j, h); l = f12_mul(k, b); m = f12_mul(k, e); n = f12_mul(s, m)
    o = f12_frob(l, 1); p = f12_mul(o, n); q = f12_frob(k, 2); r = f12_mul(q, p); t = f12_mul(f12_conj(s), l); u = f12_frob(t, 3)
    return f12_mul(u, r)

# ---- the reference's Miller loop (groups/mod.rs:486-519, 557-635), binary schedule, on affine P in E(Fq), Q in E'(Fq2): the yardstick
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531))
def ec_mul(k, P, mul, sub, inv, three_x2):
    """affine double-and-add on y^2 = x^3 + b (a = 0) over a field given by its operations"""
    def add(A, Bp):
        if A is None: return Bp
        if Bp is None: return A
        if A[0] == Bp[0]:
            if A[1] != Bp[1]: return None
            l = mul(three_x2(A[0]), inv(_dbl(A[1])))
        else:
            l = mul(sub(Bp[1], A[1]), inv(sub(Bp[0], A[0])))
        x3 = sub(sub(mul(l, l), A[0]), Bp[0])
        return (x3, sub(mul(l, sub(A[0], x3)), A[1]))
    acc = None
    while k:
        if k & 1: acc = add(acc, P)
        P = add(P, P); k >>= 1
    return acc
def _dbl(y): return f2_add(y, y) if isinstance(y, tuple) else 2 * y % Q
def g1_mul(k): return ec_mul(k, (1, 2), lambda a, b: a * b % Q, lambda a, b: (a - b) % Q, lambda a: pow(a, Q - 2, Q), lambda x: 3 * x * x % Q)
def g2_mul(k): return ec_mul(k, G2_GEN, f2_mul, f2_sub, f2_inv, lambda x: f2_scale(f2_mul(x, x), 3))
def ref_miller(P, Qa):
    xp, yp = P
    bt = f2_scale(f2_inv(XI), 3)                               # b' = 3 / xi
    half = pow(2, Q - 2, Q)
    def line(f, l0, lvw, lvv):
        return f12_mul(f, [l0, F2_ZERO, f2_scale(lvv, xp), F2_ZERO, f2_scale(lvw, yp), F2_ZERO])
    def dbl(r):
        x, y, z = r
        a = f2_scale(f2_mul(x, y), half); b = f2_mul(y, y); c = f2_mul(z, z); e = f2_mul(bt, f2_scale(c, 3)); f = f2_scale(e, 3)
        g = f2_scale(f2_add(b, f), half); yz = f2_add(y, z); h = f2_sub(f2_mul(yz, yz), f2_add(b, c)); i = f2_sub(e, b); j = f2_mul(x, x)
        r2 = (f2_mul(a, f2_sub(b, f)), f2_sub(f2_mul(g, g), f2_scale(f2_mul(e, e), 3)), f2_mul(b, h))
        return r2, (f2_mul(XI, i), f2_neg(h), f2_scale(j, 3))
    def add(r, q):
        x, y, z = r
        d = f2_sub(x, f2_mul(z, q[0])); e = f2_sub(y, f2_mul(z, q[1])); f = f2_mul(d, d); g = f2_mul(e, e); h = f2_mul(d, f); i = f2_mul(x, f)
        j = f2_sub(f2_add(f2_mul(z, g), h), f2_add(i, i))
        r2 = (f2_mul(d, j), f2_sub(f2_mul(e, f2_sub(i, j)), f2_mul(h, y)), f2_mul(z, h))
        return r2, (f2_mul(XI, f2_sub(f2_mul(e, q[0]), f2_mul(d, q[1]))), d, f2_neg(e))
    mulq = lambda q: (f2_mul(f2_conj(q[0]), gamma(1, 3, 1)), f2_mul(f2_conj(q[1]), gamma(1, 2, 1)))
    r = (Qa[0], Qa[1], F2_ONE); f = f12_one()
    n = 6 * U + 2
    for bit in bin(n)[3:]:
        f = f12_mul(f, f)
        r, (l0, lvw, lvv) = dbl(r); f = line(f, l0, lvw, lvv)
        if bit == "1":
            r, (l0, lvw, lvv) = add(r, Qa); f = line(f, l0, lvw, lvv)
    q1 = mulq(Qa); q2 = mulq(q1); q2 = (q2[0], f2_neg(q2[1]))
    r, (l0, lvw, lvv) = add(r, q1); f = line(f, l0, lvw, lvv)
    r, (l0, lvw, lvv) = add(r, q2); f = line(f, l0, lvw, lvv)
    return f

# ------------------------------------------------------------------------------------------------ register file
PAGE_DW = 576                    # one LDS page: 9 limbs x 64 slots; a register (Fq2) = two adjacent slots (even lane c0, odd lane c1)
REL = 0x8000                     # index flag: relative to the base register of the program entry
ZERO = 0
RES = list(range(1, 7))          # the running Fq12
T = list(range(8, 26))           # products of the current phase pair
V = list(range(32, 41))          # level-1 results of a product (three Fq6 values)
KBASE = {1: 41, 2: 47, 3: 53}    # Frobenius multipliers of map P: six registers each (the first one is 1)
NINV = 59                        # scratch of the easy part
NREG = 160                       # five pages
def slot(s): return 64 + 32 * (s // 3) + 9 * (s % 3)       # table slot s: 6 registers value + 3 registers -c1 (conjugate copy)
def off(reg): return 4 * ((reg >> 5) * PAGE_DW + ((reg & 31) << 1))       # BYTE offset of the register's even slot, limb 0
def rel(k): return ('rel', k)    # k-th register after the entry's base

# PROD_MULR / COMB_MR are PROD_MUL / COMB_M whose tables contain REL indices (products by a table slot, slot copies): only they pay
# for the decode of the flag
# FUSE_SQR: the recombination of one cyclotomic squaring and the products of the NEXT one in a single phase (runs of squarings)
# PROD_MUL1 / PROD_MUL2: PROD_MUL whose roles gather at most one / two registers per operand (the generic kind always gathers 4 + 4)
OPS = {"PROD_MUL": 0, "PROD_MULC": 1, "PROD_SQR": 2, "COMB_M": 3, "COMB_C": 4, "INV": 5, "COMB_M2": 6, "PROD_MULR": 7, "COMB_MR": 8, "FUSE_SQR": 9,
       "PROD_MUL1": 10, "PROD_MUL2": 11, "END": 15}

class Phase:
    def __init__(self, kind, name):
        self.kind, self.name, self.roles = kind, name, []
    def add(self, **kw):
        assert len(self.roles) < 32, self.name
        self.roles.append(kw)

def enc(r):
    if isinstance(r, tuple):
        return REL | (r[1] << 3)
    return off(r)

def role_words(kind, ro):
    """12 x uint16: PROD a[4] b[4] | COMB x[3] y[4] z pad pad ; then dst, flags (bit 0 active, bit 1 conj A / negative z)"""
    z = enc(ZERO)
    if ro is None:
        return [z] * 10 + [z, 0]
    if kind.startswith("PROD") or kind == "INV":
        a = [enc(x) for x in ro["a"]] + [z] * (4 - len(ro["a"]))
        b = [enc(x) for x in ro.get("b", [])] + [z] * (4 - len(ro.get("b", [])))
        return a + b + [z, z] + [enc(ro["dst"]), 1 | (2 if ro.get("conj") else 0)]
    if kind == "FUSE_SQR":
        pad = lambda l, n: [enc(x) for x in l] + [z] * (n - len(l))
        xs = [enc(ro.get("xp", ZERO))] + pad(ro.get("xm", []), 2)
        ys = [enc(ro.get("yp", ZERO))] + pad(ro.get("ym", []), 2)
        return xs + ys + [enc(ro.get("zp", ZERO)), enc(ro.get("zm", ZERO)), enc(ro.get("keep", ZERO)), z] + [enc(ro["dst"]), 1 | (2 if "keep" in ro else 0)]
    if kind == "COMB_M2":
        xp, xm, yp, ym = ro.get("xp", []), ro.get("xm", []), ro.get("yp", []), ro.get("ym", [])
        assert len(xp) <= 2 and len(xm) <= 2 and len(yp) <= 3 and len(ym) <= 2
        pad = lambda l, n: [enc(x) for x in l] + [z] * (n - len(l))
        return pad(xp, 2) + pad(xm, 2) + pad(yp, 3) + pad(ym, 2) + [z] + [enc(ro["dst"]), 1]
    xs = [enc(ro.get("xp", ZERO))] + [enc(x) for x in ro.get("xm", [])] + [z] * (2 - len(ro.get("xm", [])))
    ys = [enc(x) for x in ro.get("yp", [])] + [z] * (2 - len(ro.get("yp", []))) + [enc(x) for x in ro.get("ym", [])] + [z] * (2 - len(ro.get("ym", [])))
    return xs + ys + [enc(ro.get("z", ZERO)), z, z] + [enc(ro["dst"]), 1 | (2 if ro.get("zneg") else 0)]

# ------------------------------------------------------------------------------------------------ phase builders
def f6_products(ph, a, b, t0):
    """the six Karatsuba products of an Fq6 product (tower.hpp f6_mul): a, b = three LISTS of registers each (sums), -> T[t0..t0+5]
    order: aa, bb, cc, t0 = (a1+a2)(b1+b2), t1 = (a0+a1)(b0+b1), t2 = (a0+a2)(b0+b2)"""
    for k, (ia, ib) in enumerate([((0,), (0,)), ((1,), (1,)), ((2,), (2,)), ((1, 2), (1, 2)), ((0, 1), (0, 1)), ((0, 2), (0, 2))]):
        ph.add(a=sum((a[i] for i in ia), []), b=sum((b[i] for i in ib), []), dst=T[t0 + k])
def f6_combine(ph, t0, dst, negate=False):
    """c0 = xi (t0 - bb - cc) + aa ; c1 = xi cc + (t1 - aa - bb) ; c2 = t2 + bb - aa - cc   (fq6.rs:144-158)"""
    aa, bb, cc, k0, k1, k2 = (T[t0 + i] for i in range(6))
    assert not negate
    ph.add(xp=k0, xm=[bb, cc], yp=[aa], dst=dst[0])
    ph.add(xp=cc, yp=[k1], ym=[aa, bb], dst=dst[1])
    ph.add(yp=[k2, bb], ym=[aa, cc], dst=dst[2])

def mul_phases(name, a, b, out, conj_b_regs=None):
    """Fq12 product (fq12.rs:295-307 through tower.hpp f12_mul_src): a, b, out = six registers each.  Returns [PROD, COMB, COMB]."""
    p = Phase("PROD_MUL", name + ".prod")
    A0, A1 = [[a[i]] for i in range(3)], [[a[3 + i]] for i in range(3)]
    B0, B1 = [[b[i]] for i in range(3)], [[b[3 + i]] for i in range(3)]
    f6_products(p, A0, B0, 0)                                           # aa = a.c0 b.c0
    f6_products(p, A1, B1, 6)                                           # bb = a.c1 b.c1
    f6_products(p, [A0[i] + A1[i] for i in range(3)], [B0[i] + B1[i] for i in range(3)], 12)     # t = (a.c0 + a.c1)(b.c0 + b.c1)
    l1 = Phase("COMB_M", name + ".l1")
    f6_combine(l1, 0, V[0:3]); f6_combine(l1, 6, V[3:6]); f6_combine(l1, 12, V[6:9])
    l2 = Phase("COMB_M", name + ".l2")
    aa, bb, t = V[0:3], V[3:6], V[6:9]
    l2.add(xp=bb[2], yp=[aa[0]], dst=out[0])                            # aa + v bb
    l2.add(yp=[aa[1], bb[0]], dst=out[1])
    l2.add(yp=[aa[2], bb[1]], dst=out[2])
    for k in range(3):
        l2.add(yp=[t[k]], ym=[aa[k], bb[k]], dst=out[3 + k])            # t - aa - bb
    return [p, l1, l2]

def cyc_phases():
    """Granger-Scott cyclotomic squaring (fq12.rs:178-227) with three Fq2 SQUARINGS per Fp4 (a^2, b^2, (a+b)^2)"""
    z = {0: RES[0], 4: RES[1], 3: RES[2], 2: RES[3], 1: RES[4], 5: RES[5]}
    p = Phase("PROD_SQR", "cyc.prod")
    for g, (ia, ib) in enumerate([(0, 1), (2, 3), (4, 5)]):
        p.add(a=[z[ia]], dst=T[3 * g]); p.add(a=[z[ib]], dst=T[3 * g + 1]); p.add(a=[z[ia], z[ib]], dst=T[3 * g + 2])
    c = Phase("COMB_C", "cyc.comb")
    def even(g, zi): c.add(xp=T[3 * g + 1], yp=[T[3 * g]], z=z[zi], zneg=True, dst=z[zi])                           # 3 (a^2 + xi b^2) - 2 z
    def odd(g, zi, xi_):
        if xi_: c.add(xp=T[3 * g + 2], xm=[T[3 * g], T[3 * g + 1]], z=z[zi], dst=z[zi])                             # 3 xi (2ab) + 2 z
        else: c.add(yp=[T[3 * g + 2]], ym=[T[3 * g], T[3 * g + 1]], z=z[zi], dst=z[zi])                            # 3 (2ab) + 2 z
    even(0, 0); odd(0, 1, False); even(1, 4); odd(1, 5, False); even(2, 3); odd(2, 2, True)
    return [p, c]

RES1 = list(range(26, 32))         # second bank of the running value (runs of fused squarings alternate between the two)
def cyc_fused_phases():
    """Runs of cyclotomic squarings: after the first products, ONE phase per further squaring - every active pair reduces the operand
    of its NEXT square from the previous products (the Granger-Scott update is linear, so the (a + b) pairs reduce a' + b' directly:
    3 xi X + 3 Y + 2 (z_b - z_a)) and squares it.  Products and running value alternate between two register banks, so no pair reads
    what another writes.  Returns (fuse[parity], final_comb[parity]): fuse[p] reads bank p and writes bank 1 - p; final_comb[p]
    turns the products of bank p into the running value in bank 0."""
    zi = {0: 0, 4: 1, 3: 2, 2: 3, 1: 4, 5: 5}                                    # z index -> position in RES
    bank = lambda p: (T[9 * p: 9 * p + 9], RES if p == 0 else RES1)
    fuse, final = {}, {}
    for p in (0, 1):
        (S, Z), (S2, Z2) = bank(p), bank(1 - p)
        z = lambda i: Z[zi[i]]
        zn = lambda i: Z2[zi[i]]
        f = Phase("FUSE_SQR", "cycf%d" % p)
        # group 0 (a, b) = (z0, z1): next operands from its own products S[0..2]
        f.add(xp=S[1], yp=S[0], zm=z(0), keep=zn(0), dst=S2[0])
        f.add(yp=S[2], ym=[S[0], S[1]], zp=z(1), keep=zn(1), dst=S2[1])
        f.add(xp=S[1], yp=S[2], ym=[S[1]], zp=z(1), zm=z(0), dst=S2[2])
        # group 1 (a, b) = (z2, z3): next operands z2', z3' come from the products of group 2, S[6..8]
        f.add(xp=S[8], xm=[S[6], S[7]], zp=z(2), keep=zn(2), dst=S2[3])
        f.add(xp=S[7], yp=S[6], zm=z(3), keep=zn(3), dst=S2[4])
        f.add(xp=S[8], xm=[S[6]], yp=S[6], zp=z(2), zm=z(3), dst=S2[5])
        # group 2 (a, b) = (z4, z5): next operands from the products of group 1, S[3..5]
        f.add(xp=S[4], yp=S[3], zm=z(4), keep=zn(4), dst=S2[6])
        f.add(yp=S[5], ym=[S[3], S[4]], zp=z(5), keep=zn(5), dst=S2[7])
        f.add(xp=S[4], yp=S[5], ym=[S[4]], zp=z(5), zm=z(4), dst=S2[8])
        fuse[p] = f
        c = Phase("COMB_C", "cyc.final%d" % p)
        out = lambda i: RES[zi[i]]
        def even(g, i): c.add(xp=S[3 * g + 1], yp=[S[3 * g]], z=z(i), zneg=True, dst=out(i))
        def odd(g, i, xi_):
            if xi_: c.add(xp=S[3 * g + 2], xm=[S[3 * g], S[3 * g + 1]], z=z(i), dst=out(i))
            else: c.add(yp=[S[3 * g + 2]], ym=[S[3 * g], S[3 * g + 1]], z=z(i), dst=out(i))
        even(0, 0); odd(0, 1, False); even(1, 4); odd(1, 5, False); even(2, 3); odd(2, 2, True)
        final[p] = c
    return fuse, final

def copy_phase(name, src, dst, neg=()):
    """dst[i] = src[i] (value preserving reduction), or -src[i] for i in neg"""
    c = Phase("COMB_M", name)
    for i, (s, d) in enumerate(zip(src, dst)):
        if i in neg: c.add(ym=[s], dst=d)
        else: c.add(yp=[s], dst=d)
    return c

def frob_phase(P):
    """fq12.rs:90-95 with the three constant products per coefficient folded into one (KBASE[P] holds them): ONE product per pair"""
    p = Phase("PROD_MULC" if P & 1 else "PROD_MUL", "frob%d" % P)
    for j in range(6):
        p.add(a=[RES[j]], b=[KBASE[P] + j], dst=RES[j], conj=bool(P & 1))
    return p
def frob_constants(P):
    one = F2_ONE
    return [one, FROB6_C1[P], FROB6_C2[P], FROB12_C1[P], f2_mul(FROB6_C1[P], FROB12_C1[P]), f2_mul(FROB6_C2[P], FROB12_C1[P])]

# ------------------------------------------------------------------------------------------------ the Miller loop as a program
# Registers of the Miller loop live in pages 2-4 (the exponentiation's table slots, which only the final exponentiation uses).
_m = iter(range(64, 160))
def _take(n=1):
    r = [next(_m) for _ in range(n)]
    return r if n > 1 else r[0]
TR = _take(9)                      # products of the running point in the first phase of a doubling step
TA = _take(10)                     # products / temporaries of an addition step
MH, ML0, MLVW, MLVV = _take(4)     # h and the line (ell_0, ell_vw * yP, ell_vv * xP)
RX, RY, RZ, ZB, NZB = _take(5)     # R (homogeneous projective), and +-(27 - 3i) z
QX, QY, NQY, PX, NPX, PY, NPY = _take(7)
MD, ME, MJ, MT = _take(4)
C_ONE, C_T2, C_NT2, C_T3, C_NT3, C_B3, C_NB3, C_TWX, C_TWY, C_NTWY = _take(10)      # constants, written by the kernel's prologue
IN_PX, IN_PY, IN_PZ, IN_QX, IN_QY, IN_QZ = _take(6)                                 # inputs: P as (x, 0) ..., Q
IZP, IZQ, IZP2, IZQ2, IZP3, IZQ3, PXA, PYA, QXA, QYA = _take(10)
T27 = list(range(8, 26))           # the 18 products of f * f (the T area of the product)

def iso_constants():
    """t with t^6 = 82/3: the curve constant 3 b' t^6 of the isomorphic twist is 27 - 3i (gen_device_constants.py ISO_T2 / ISO_T3)"""
    consts = (pathlib.Path(__file__).resolve().parents[1] / "bn_amd" / "csrc" / "bn254_constants.hpp").read_text()
    import re
    def limbs(name):
        v = [int(x, 16) for x in re.search(name + r"\[9\] = \{([^}]*)\}", consts).group(1).replace("u", "").split(",")]
        m = sum(l << (29 * i) for i, l in enumerate(v))
        return m * pow(1 << 261, -1, Q) % Q
    t2, t3 = limbs("ISO_T2"), limbs("ISO_T3")
    assert pow(t2, 3, Q) == 82 * pow(3, -1, Q) % Q and t3 * t3 % Q == pow(t2, 3, Q)
    return t2, t3
def miller_constants():
    t2, t3 = iso_constants()
    gx, gy = gamma(1, 3, 1), gamma(1, 2, 1)
    return {C_ONE: F2_ONE, C_T2: (t2, 0), C_NT2: (Q - t2, 0), C_T3: (t3, 0), C_NT3: (Q - t3, 0), C_B3: (27, Q - 3), C_NB3: (Q - 27, 3),
            C_TWX: gx, C_TWY: gy, C_NTWY: f2_neg(gy)}

def ate_naf():
    n, naf = 6 * U + 2, []
    while n:
        z = (2 - (n % 4)) if (n & 1) else 0
        n -= z; naf.append(z); n //= 2
    return naf

def miller_program(B):
    """the fused NAF Miller loop of pairing.hpp miller_loop_sched<true> (groups/mod.rs:486-519 + 557-635 on the isomorphic curve),
    division-free: the doubling step keeps 4R instead of R (homogeneous coordinates), which scales later lines by elements of Fq
    that the final exponentiation removes.  Six phases per doubling step, seven per addition step."""
    E = lambda ph: B.entry(ph)
    prog = []
    # ---- prologue: affine P and Q (one INV phase for both z), onto the isomorphic curve, R = Q, f = 1
    ph = Phase("INV", "m.inv"); ph.add(a=[IN_PZ], dst=IZP); ph.add(a=[IN_QZ], dst=IZQ); prog.append(E(ph))
    ph = Phase("PROD_MUL", "m.p1"); ph.add(a=[IZP], b=[IZP], dst=IZP2); ph.add(a=[IZQ], b=[IZQ], dst=IZQ2); prog.append(E(ph))
    ph = Phase("PROD_MUL", "m.p2")
    ph.add(a=[IZP2], b=[IZP], dst=IZP3); ph.add(a=[IZQ2], b=[IZQ], dst=IZQ3); ph.add(a=[IN_PX], b=[IZP2], dst=PXA); ph.add(a=[IN_QX], b=[IZQ2], dst=QXA)
    prog.append(E(ph))
    ph = Phase("PROD_MUL", "m.p3")
    ph.add(a=[IN_PY], b=[IZP3], dst=PYA); ph.add(a=[IN_QY], b=[IZQ3], dst=QYA)
    ph.add(a=[PXA], b=[C_T2], dst=PX); ph.add(a=[PXA], b=[C_NT2], dst=NPX); ph.add(a=[QXA], b=[C_T2], dst=QX); ph.add(a=[QXA], b=[C_T2], dst=RX)
    prog.append(E(ph))
    ph = Phase("PROD_MUL", "m.p4")
    ph.add(a=[PYA], b=[C_T3], dst=PY); ph.add(a=[PYA], b=[C_NT3], dst=NPY); ph.add(a=[QYA], b=[C_T3], dst=QY); ph.add(a=[QYA], b=[C_NT3], dst=NQY)
    ph.add(a=[QYA], b=[C_T3], dst=RY); ph.add(a=[C_ONE], b=[C_ONE], dst=RZ); ph.add(a=[C_ONE], b=[C_B3], dst=ZB); ph.add(a=[C_ONE], b=[C_NB3], dst=NZB)
    ph.add(a=[C_ONE], b=[C_ONE], dst=RES[0])
    for k in range(1, 6): ph.add(a=[ZERO], b=[ZERO], dst=RES[k])
    prog.append(E(ph))
    # ---- the sparse product f * (l0 + lvv v^2 + lvw v w) (fq12.rs:107-176 as a schoolbook product: 18 products, every output 3 terms)
    a = RES
    def sparse_products(ph):
        for k in range(6):
            ph.add(a=[a[k]], b=[ML0], dst=T27[3 * k]); ph.add(a=[a[k]], b=[MLVV], dst=T27[3 * k + 1]); ph.add(a=[a[k]], b=[MLVW], dst=T27[3 * k + 2])
    def sparse_combine(ph):
        t = lambda k, w: T27[3 * k + {"l0": 0, "lvv": 1, "lvw": 2}[w]]
        ph.add(xp=[t(1, "lvv"), t(4, "lvw")], yp=[t(0, "l0")], dst=a[0])                       # a00 l0 + xi (a01 lvv + a11 lvw)
        ph.add(xp=[t(2, "lvv"), t(5, "lvw")], yp=[t(1, "l0")], dst=a[1])
        ph.add(yp=[t(2, "l0"), t(0, "lvv"), t(3, "lvw")], dst=a[2])
        ph.add(xp=[t(2, "lvw"), t(4, "lvv")], yp=[t(3, "l0")], dst=a[3])
        ph.add(xp=[t(5, "lvv")], yp=[t(0, "lvw"), t(4, "l0")], dst=a[4])
        ph.add(yp=[t(1, "lvw"), t(5, "l0"), t(3, "lvv")], dst=a[5])
    # ---- doubling step
    XY, TB, TC, TS, TJ, TE, TNE, YZB, YNZB = TR
    d1 = Phase("PROD_MUL", "m.d1")
    A0, A1 = [[a[i]] for i in range(3)], [[a[3 + i]] for i in range(3)]
    f6_products(d1, A0, A0, 0); f6_products(d1, A1, A1, 6); f6_products(d1, [A0[i] + A1[i] for i in range(3)], [A0[i] + A1[i] for i in range(3)], 12)
    for (x, y, dst) in [([RX], [RY], XY), ([RY], [RY], TB), ([RZ], [RZ], TC), ([RY, RZ], [RY, RZ], TS), ([RX], [RX], TJ), ([RZ], [ZB], TE), ([RZ], [NZB], TNE),
                        ([RY], [ZB], YZB), ([RY], [NZB], YNZB)]:
        d1.add(a=x, b=y, dst=dst)
    d2 = Phase("COMB_M2", "m.d2")
    def f6c(ph, t0, dst):
        aa, bb, cc, k0, k1, k2 = (T[t0 + i] for i in range(6))
        ph.add(xp=[k0], xm=[bb, cc], yp=[aa], dst=dst[0]); ph.add(xp=[cc], yp=[k1], ym=[aa, bb], dst=dst[1]); ph.add(yp=[k2, bb], ym=[aa, cc], dst=dst[2])
    f6c(d2, 0, V[0:3]); f6c(d2, 6, V[3:6]); f6c(d2, 12, V[6:9])
    d2.add(yp=[TS], ym=[TB, TC], dst=MH)                                                        # h = (y + z)^2 - b - c
    d3 = Phase("COMB_M2", "m.d3")
    aa, bb, tt = V[0:3], V[3:6], V[6:9]
    d3.add(xp=[bb[2]], yp=[aa[0]], dst=a[0]); d3.add(yp=[aa[1], bb[0]], dst=a[1]); d3.add(yp=[aa[2], bb[1]], dst=a[2])
    for k in range(3): d3.add(yp=[tt[k]], ym=[aa[k], bb[k]], dst=a[3 + k])
    d3.add(xp=[TE], xm=[TB], dst=ML0)                                                           # ell_0 = xi (e - b)
    d4 = Phase("PROD_MUL", "m.d4")
    TG, TE12 = TA[0], TA[1]
    d4.add(a=[XY, XY], b=[TB, TNE, TNE, TNE], dst=RX)                                          # 4 x' = 2xy (b - 3e)
    d4.add(a=[TB, TE, TE, TE], b=[TB, TE, TE, TE], dst=TG)                                     # (b + 3e)^2
    d4.add(a=[TE, TE, TE, TE], b=[TE, TE, TE], dst=TE12)                                       # 12 e^2
    d4.add(a=[TB, TB, TB, TB], b=[MH], dst=RZ)                                                 # 4 z' = 4 b h
    d4.add(a=[TB, TB, TB, TB], b=[YZB, YZB], dst=ZB); d4.add(a=[TB, TB, TB, TB], b=[YNZB, YNZB], dst=NZB)     # +-(27 - 3i) z' = 4b * 2y * (+-(27 - 3i) z)
    d4.add(a=[MH], b=[NPY], dst=MLVW); d4.add(a=[TJ, TJ, TJ], b=[PX], dst=MLVV)                # ell_vw yP = -h yP ; ell_vv xP = 3 x^2 xP
    d5 = Phase("PROD_MUL", "m.d5"); sparse_products(d5)
    d6 = Phase("COMB_M2", "m.d6"); sparse_combine(d6)
    d6.add(yp=[TG], ym=[TE12], dst=RY)                                                          # 4 y' = (b + 3e)^2 - 12 e^2
    DBL = [E(x) for x in (d1, d2, d3, d4, d5, d6)]
    # ---- addition step R += (qx, qy)  (groups/mod.rs:592-610)
    def add_step(qx, qy):
        ZQX, ZQY, TF, TGG, EQX, DQY, TH, TI, TI2, TZG = TA
        a1 = Phase("PROD_MUL", "m.a1"); a1.add(a=[RZ], b=[qx], dst=ZQX); a1.add(a=[RZ], b=[qy], dst=ZQY)
        a2 = Phase("COMB_M2", "m.a2"); a2.add(yp=[RX], ym=[ZQX], dst=MD); a2.add(yp=[RY], ym=[ZQY], dst=ME)
        a3 = Phase("PROD_MUL", "m.a3")
        a3.add(a=[MD], b=[MD], dst=TF); a3.add(a=[ME], b=[ME], dst=TGG); a3.add(a=[ME], b=[qx], dst=EQX); a3.add(a=[MD], b=[qy], dst=DQY)
        a3.add(a=[ME], b=[NPX], dst=MLVV); a3.add(a=[MD], b=[PY], dst=MLVW)                     # ell_vv xP = -e xP ; ell_vw yP = d yP
        a4 = Phase("PROD_MUL", "m.a4")
        a4.add(a=[MD], b=[TF], dst=TH); a4.add(a=[RX], b=[TF], dst=TI); a4.add(a=[RX, RX], b=[TF], dst=TI2); a4.add(a=[RZ], b=[TGG], dst=TZG)
        a5 = Phase("COMB_M2", "m.a5")
        a5.add(yp=[TZG, TH], ym=[TI2], dst=MJ)                                                  # j = z g + h - 2 i
        a5.add(yp=[TI, TI2], ym=[TZG, TH], dst=MT)                                              # i - j
        a5.add(xp=[EQX], xm=[DQY], dst=ML0)                                                     # ell_0 = xi (e qx - d qy)
        a6 = Phase("PROD_MUL", "m.a6"); sparse_products(a6)
        ET, HY = TA[0], TA[1]                                                                   # (ZQX, ZQY are dead)
        a6.add(a=[ME], b=[MT], dst=ET); a6.add(a=[TH], b=[RY], dst=HY); a6.add(a=[MD], b=[MJ], dst=RX); a6.add(a=[RZ], b=[TH], dst=RZ)
        a6.add(a=[ZB], b=[TH], dst=ZB); a6.add(a=[NZB], b=[TH], dst=NZB)                         # (27 - 3i) z' = ((27 - 3i) z) h
        a7 = Phase("COMB_M2", "m.a7"); sparse_combine(a7); a7.add(yp=[ET], ym=[HY], dst=RY)
        return [E(x) for x in (a1, a2, a3, a4, a5, a6, a7)]
    ADD_P, ADD_N = add_step(QX, QY), add_step(QX, NQY)
    naf = ate_naf()
    nd = len(naf) - 1
    for jj in range(nd):
        prog += DBL
        dgt = naf[nd - 1 - jj]
        if dgt: prog += ADD_P if dgt > 0 else ADD_N
    # ---- pi(Q), then -pi^2(Q)   (groups/mod.rs:578-582; mul_by_q :550-555 commutes with the isomorphism)
    Q1X, Q1Y, Q2X, Q2Y = IZP, IZQ, IZP2, IZQ2                       # (prologue registers, dead by now)
    f1 = Phase("PROD_MULC", "m.q1"); f1.add(a=[QX], b=[C_TWX], dst=Q1X, conj=True); f1.add(a=[QY], b=[C_TWY], dst=Q1Y, conj=True)
    f2 = Phase("PROD_MULC", "m.q2"); f2.add(a=[Q1X], b=[C_TWX], dst=Q2X, conj=True); f2.add(a=[Q1Y], b=[C_NTWY], dst=Q2Y, conj=True)
    prog += [E(f1)] + add_step(Q1X, Q1Y) + [E(f2)] + add_step(Q2X, Q2Y)
    return prog

# ------------------------------------------------------------------------------------------------ programs
class Builder:
    def __init__(self):
        self.phases, self.index = [], {}
    def pid(self, ph):
        key = (ph.kind, tuple(tuple(role_words(ph.kind, r)) for r in ph.roles), ph.name if ph.name.startswith("mulr") else None)
        if key not in self.index:
            self.index[key] = len(self.phases); self.phases.append(ph)
        return self.index[key]
    def entry(self, ph, base=0):
        uses_rel = any(w & REL for r in ph.roles for w in role_words(ph.kind, r)[:11])
        kind = ph.kind
        if uses_rel:
            kind = {"PROD_MUL": "PROD_MULR", "COMB_M": "COMB_MR"}[kind]
        else:
            base = 0                                             # no relative index in this table: the base is never read
            if kind == "PROD_MUL":
                terms = max(max(len(r["a"]), len(r.get("b", []))) for r in ph.roles)
                kind = "PROD_MUL1" if terms <= 1 else "PROD_MUL2" if terms <= 2 else "PROD_MUL"
        return (OPS[kind], self.pid(ph), off(base))

def build():
    B = Builder()
    progs = {}
    relv = [rel(k) for k in range(6)]
    relc = [rel(0), rel(1), rel(2), rel(6), rel(7), rel(8)]               # the conjugate copy of a slot: (c0, -c1)
    MUL = mul_phases("mul", RES, relv, RES)
    MULC = mul_phases("mulc", RES, relc, RES)
    CYC = cyc_phases()
    PUT = copy_phase("put", RES + RES[3:], [rel(k) for k in range(9)], neg=(6, 7, 8))
    GET = copy_phase("get", relv, RES)
    CONJ = copy_phase("conj", RES[3:], RES[3:], neg=(0, 1, 2))
    FROB = {P: frob_phase(P) for P in (1, 2, 3)}
    def mul_by(s, conj=False): return [B.entry(ph, slot(s)) for ph in (MULC if conj else MUL)]
    def cyc(): return [B.entry(ph) for ph in CYC]
    FUSE, FINAL = cyc_fused_phases()
    def fuse_runs(prog):
        """k >= 2 consecutive squarings [P C] [P C] ... -> P F F ... F C': k + 1 phases instead of 2k"""
        one = cyc(); out = []; i = 0
        while i < len(prog):
            k = 0
            while prog[i + 2 * k: i + 2 * k + 2] == one: k += 1
            if k >= 2:
                out.append(one[0])
                out += [B.entry(FUSE[j % 2]) for j in range(k - 1)]
                out.append(B.entry(FINAL[(k - 1) % 2]))
                i += 2 * k
            else:
                out.append(prog[i]); i += 1
        return out
    def put(s): return [B.entry(PUT, slot(s))]
    def get(s): return [B.entry(GET, slot(s))]
    END = [(OPS["END"], 0, 0)]
    # ---- unit programs (tests, product tail)
    progs["MUL"] = mul_by(0) + END                                        # RES <- RES * slot 0
    progs["MULC"] = mul_by(0, True) + END                                 # RES <- RES * conj(slot 0)   (slot written by PUT)
    progs["CYC"] = cyc() + END
    progs["PUT0"] = put(0) + END
    for P in (1, 2, 3): progs["FROB%d" % P] = [B.entry(FROB[P])] + END
    # the product used by the in-launch product tree (bn254_gt_reduce_W): RES <- RES * (the six registers at KBASE[1], which that
    # kernel uses as its operand area), so that the whole machine fits in two LDS pages; its three tables are consecutive
    MULR = mul_phases("mulr", RES, [KBASE[1] + k for k in range(6)], RES)
    progs["MULR"] = [B.entry(ph) for ph in MULR] + END
    B.mulr0 = B.pid(MULR[0])
    assert [B.pid(ph) for ph in MULR] == [B.mulr0, B.mulr0 + 1, B.mulr0 + 2]
    # ---- easy part (fq12.rs:41-52): f in RES -> s in RES
    e = []
    SF, SA, SB = 0, 1, 2                                                  # slots: f, scratch, scratch (all rewritten by the hard part)
    e += put(SF)
    f = [slot(SF) + k for k in range(6)]
    # s0 = f.c0^2, s1 = f.c1^2 (schoolbook, doubled operands), d = s0 - v s1
    p = Phase("PROD_MUL", "inv.sq")
    for h in (0, 3):
        a0, a1, a2 = f[h], f[h + 1], f[h + 2]
        for (x, y) in [([a0], [a0]), ([a1, a1], [a2]), ([a0, a0], [a1]), ([a2], [a2]), ([a0, a0], [a2]), ([a1], [a1])]:
            p.add(a=x, b=y, dst=T[len(p.roles)])
    P_, Q_ = T[0:6], T[6:12]
    D = [slot(SA) + k for k in range(3)]; C = [slot(SA) + 3 + k for k in range(3)]; TI = [slot(SB) + k for k in range(3)]
    c = Phase("COMB_M", "inv.d")
    c.add(xp=P_[1], xm=[Q_[4], Q_[5]], yp=[P_[0]], dst=D[0])
    c.add(xp=P_[3], xm=[Q_[1]], yp=[P_[2]], ym=[Q_[0]], dst=D[1])
    c.add(xm=[Q_[3]], yp=[P_[4], P_[5]], ym=[Q_[2]], dst=D[2])
    e += [B.entry(p), B.entry(c)]
    # f6_inverse(d) (fq6.rs:129-141)
    p = Phase("PROD_MUL", "inv.f6a")
    for (x, y) in [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]:
        p.add(a=[D[x]], b=[D[y]], dst=T[len(p.roles)])
    c = Phase("COMB_M", "inv.f6b")
    c.add(xm=[T[5]], yp=[T[0]], dst=C[0]); c.add(xp=T[2], ym=[T[3]], dst=C[1]); c.add(yp=[T[1]], ym=[T[4]], dst=C[2])
    e += [B.entry(p), B.entry(c)]
    p = Phase("PROD_MUL", "inv.n")
    p.add(a=[D[2]], b=[C[1]], dst=T[0]); p.add(a=[D[1]], b=[C[2]], dst=T[1]); p.add(a=[D[0]], b=[C[0]], dst=T[2])
    c1 = Phase("COMB_M", "inv.n1"); c1.add(xp=T[0], yp=[T[2]], dst=T[3])
    c2 = Phase("COMB_M", "inv.n2"); c2.add(xp=T[1], yp=[T[3]], dst=T[4])
    iv = Phase("INV", "inv.f2"); iv.add(a=[T[4]], dst=NINV)
    p2 = Phase("PROD_MUL", "inv.t")
    for k in range(3): p2.add(a=[NINV], b=[C[k]], dst=TI[k])
    e += [B.entry(p), B.entry(c1), B.entry(c2), B.entry(iv), B.entry(p2)]
    # b' = conj(f^-1) = (f.c0 t, f.c1 t): two Fq6 products; then f b' = conj(conj(f) f^-1)
    p = Phase("PROD_MUL", "inv.b")
    f6_products(p, [[f[0]], [f[1]], [f[2]]], [[TI[0]], [TI[1]], [TI[2]]], 0)
    f6_products(p, [[f[3]], [f[4]], [f[5]]], [[TI[0]], [TI[1]], [TI[2]]], 6)
    BP = [slot(SB) + 3 + k for k in range(6)]
    c = Phase("COMB_M", "inv.bc"); f6_combine(c, 0, BP[0:3]); f6_combine(c, 6, BP[3:6])
    e += [B.entry(p), B.entry(c)]
    e += get(SF)
    e += [B.entry(ph) for ph in mul_phases("inv.m", RES, BP, RES)] + [B.entry(CONJ)]            # c = conj(f b')
    e += put(SA) + [B.entry(FROB[2])] + mul_by(SA)                                                # frob2(c) c
    progs["EASY"] = e + END
    # ---- hard part: the engine's own program (tools/gen_device_constants.py FE_PROG, pairing.hpp fe_step), re-expanded
    consts = (pathlib.Path(__file__).resolve().parents[1] / "bn_amd" / "csrc" / "bn254_constants.hpp").read_text()
    import re
    fe_prog = [int(x) for x in re.search(r"FE_PROG\[\d+\] = \{([^}]*)\}", consts).group(1).split(",")]
    h = []
    for w in fe_prog:
        g, m, pt, post = (w >> 10) & 15, (w >> 1) & 15, (w >> 6) & 15, (w >> 14) & 7
        if g: h += get(g - 1)
        if w & 1: h += cyc()
        if m: h += mul_by(m - 1, bool((w >> 5) & 1))
        if post == 1: h += [B.entry(CONJ)]
        elif post: h += [B.entry(FROB[post - 1])]
        if pt: h += put(pt - 1)
    progs["CYC5"] = fuse_runs(cyc() * 5) + END                            # five squarings as one fused run (tests, measurement)
    h = fuse_runs(h)
    progs["HARD"] = h + END
    progs["FE"] = e + h + END
    B.nfe = len(B.phases)                                                   # the tables of the final exponentiation come first
    mil = miller_program(B)
    progs["MILLER"] = mil + END
    progs["PAIRING"] = mil + e + h + END
    return B, progs

# ------------------------------------------------------------------------------------------------ executor on exact values
def run(B, prog, regs):
    def rd(i, base):
        if isinstance(i, tuple): return regs[base + i[1]]
        return regs[i]
    def wr_idx(i, base): return base + i[1] if isinstance(i, tuple) else i
    boff = {off(r): r for r in range(NREG)}
    for (op, pid, b) in prog:
        if op == OPS["END"]: break
        ph = B.phases[pid]; base = boff[b]
        reads, writes = {}, {}
        out = []
        for pi, ro in enumerate(ph.roles):
            def note(i):
                reads.setdefault(wr_idx(i, base), set()).add(pi)
                return rd(i, base)
            if ph.kind.startswith("PROD") or ph.kind == "INV":
                A = F2_ZERO
                for i in ro["a"]: A = f2_add(A, note(i))
                if ro.get("conj"): A = f2_conj(A)
                if ph.kind == "PROD_SQR": val = f2_mul(A, A)
                elif ph.kind == "INV": val = f2_inv(A)
                else:
                    Bv = F2_ZERO
                    for i in ro["b"]: Bv = f2_add(Bv, note(i))
                    val = f2_mul(A, Bv)
            elif ph.kind == "FUSE_SQR":
                X = note(ro["xp"]) if "xp" in ro else F2_ZERO
                for i in ro.get("xm", []): X = f2_sub(X, note(i))
                Y = note(ro["yp"]) if "yp" in ro else F2_ZERO
                for i in ro.get("ym", []): Y = f2_sub(Y, note(i))
                Z = f2_sub(note(ro["zp"]) if "zp" in ro else F2_ZERO, note(ro["zm"]) if "zm" in ro else F2_ZERO)
                opnd = f2_add(f2_scale(f2_add(f2_mul(XI, X), Y), 3), f2_scale(Z, 2))
                if "keep" in ro:
                    kd = wr_idx(ro["keep"], base)
                    assert kd not in writes; writes[kd] = pi; out.append((kd, opnd))
                val = f2_mul(opnd, opnd)
            elif ph.kind == "COMB_M2":
                X = F2_ZERO; Y = F2_ZERO
                for i in ro.get("xp", []): X = f2_add(X, note(i))
                for i in ro.get("xm", []): X = f2_sub(X, note(i))
                for i in ro.get("yp", []): Y = f2_add(Y, note(i))
                for i in ro.get("ym", []): Y = f2_sub(Y, note(i))
                val = f2_add(f2_mul(XI, X), Y)
            else:
                X = note(ro["xp"]) if "xp" in ro else F2_ZERO
                for i in ro.get("xm", []): X = f2_sub(X, note(i))
                Y = F2_ZERO
                for i in ro.get("yp", []): Y = f2_add(Y, note(i))
                for i in ro.get("ym", []): Y = f2_sub(Y, note(i))
                if ph.kind == "COMB_M": val = f2_add(f2_mul(XI, X), Y)
                else:
                    Z = note(ro["z"])
                    val = f2_add(f2_scale(f2_add(f2_mul(XI, X), Y), 3), f2_scale(Z, -2 if ro.get("zneg") else 2))
            d = wr_idx(ro["dst"], base)
            assert d not in writes, "two pairs write one register in " + ph.name
            writes[d] = pi; out.append((d, val))
        for d, pi in writes.items():                       # hazard rule: a register written in a phase is read by its writer only
            assert reads.get(d, set()) <= {pi}, "phase %s: register %d is read by another pair while written" % (ph.name, d)
        for d, val in out: regs[d] = val
    return regs

def fresh_regs():
    regs = [F2_ZERO] * NREG
    for P in (1, 2, 3):
        for j, cst in enumerate(frob_constants(P)): regs[KBASE[P] + j] = cst
    for r, v in miller_constants().items(): regs[r] = v
    return regs

def self_check(B, progs):
    rnd = random.Random(7)
    rf2 = lambda: (rnd.randrange(Q), rnd.randrange(Q))
    rf12 = lambda: [rf2() for _ in range(6)]
    def with_res(x):
        regs = fresh_regs()
        for r, v in zip(RES, x): regs[r] = v
        return regs
    a, b = rf12(), rf12()
    regs = with_res(b); run(B, progs["PUT0"], regs)
    for r, v in zip(RES, a): regs[r] = v
    r1 = list(regs); run(B, progs["MUL"], r1)
    assert [r1[r] for r in RES] == f12_mul(a, b), "MUL"
    r1 = list(regs); run(B, progs["MULC"], r1)
    assert [r1[r] for r in RES] == f12_mul(a, f12_conj(b)), "MULC"
    regs = with_res(a)
    for k in range(6): regs[KBASE[1] + k] = b[k]
    run(B, progs["MULR"], regs)
    assert [regs[r] for r in RES] == f12_mul(a, b), "MULR"
    for P in (1, 2, 3):
        regs = with_res(a); run(B, progs["FROB%d" % P], regs)
        assert [regs[r] for r in RES] == f12_frob(a, P), "FROB"
    # a cyclotomic element: first chunk of the final exponentiation of a random value
    c = f12_mul(f12_conj(a), f12_inv(a)); cyc_el = f12_mul(f12_frob(c, 2), c)
    regs = with_res(cyc_el); run(B, progs["CYC"], regs)
    assert [regs[r] for r in RES] == f12_mul(cyc_el, cyc_el), "CYC"
    regs = with_res(cyc_el); run(B, progs["CYC5"], regs)
    assert [regs[r] for r in RES] == f12_pow(cyc_el, 32), "CYC5"
    regs = with_res(a); run(B, progs["EASY"], regs)
    assert [regs[r] for r in RES] == cyc_el, "EASY"
    regs = with_res(a); run(B, progs["FE"], regs)
    assert [regs[r] for r in RES] == final_exponentiation(a), "FE"
    # the Miller loop: Jacobian inputs with z != 1; the machine's NAF / isomorphic-curve / division-free value differs from the
    # reference's by factors the final exponentiation removes, so the comparison is after it - i.e. pairing() itself
    P, Qa = g1_mul(rnd.randrange(R_ORD)), g2_mul(rnd.randrange(R_ORD))
    want = final_exponentiation(ref_miller(P, Qa))
    zp, zq = rnd.randrange(1, Q), rf2()
    regs = fresh_regs()
    regs[IN_PX], regs[IN_PY], regs[IN_PZ] = (P[0] * zp * zp % Q, 0), (P[1] * pow(zp, 3, Q) % Q, 0), (zp, 0)
    zq2 = f2_mul(zq, zq)
    regs[IN_QX], regs[IN_QY], regs[IN_QZ] = f2_mul(Qa[0], zq2), f2_mul(Qa[1], f2_mul(zq2, zq)), zq
    r1 = list(regs); run(B, progs["MILLER"], r1)
    assert final_exponentiation([r1[r] for r in RES]) == want, "MILLER"
    run(B, progs["PAIRING"], regs)
    assert [regs[r] for r in RES] == want, "PAIRING"

# ------------------------------------------------------------------------------------------------ emit
def mont_limbs(a):
    v = a % Q * (1 << 261) % Q
    return "{" + ", ".join("0x%08xu" % ((v >> (29 * i)) & ((1 << 29) - 1)) for i in range(9)) + "}"

def emit(B, progs, path):
    o = ["// GENERATED by tools/gen_wave_tables.py - do not edit.  Role tables and programs of the wave-cooperative Fq12 machine (wave.hpp).",
         "#pragma once", "#include <stdint.h>", "#ifndef BN254_CONSTANT", "#define BN254_CONSTANT constexpr", "#endif",
         "namespace bn254 { namespace wv {",
         "constexpr int PAGE_DW = %d, NREG = %d, NPAGES = %d, REL = 0x%x;" % (PAGE_DW, NREG, NREG // 32, REL),
         "constexpr int OFF_ZERO = %d, OFF_RES = %d, OFF_SLOT0 = %d, OFF_NINV = %d;" % (off(ZERO), off(RES[0]), off(slot(0)), off(NINV))]
    o.append("enum { " + ", ".join("OP_%s = %d" % kv for kv in OPS.items()) + " };")
    o.append("constexpr int KBASE_OFF[4] = {0, %d, %d, %d};          // Frobenius multipliers of map P: six registers from here" % tuple(off(KBASE[P]) for P in (1, 2, 3)))
    o.append("// the Frobenius multipliers as the engine's 9 x 29-bit Montgomery limbs (radix 2^261): [3 maps x 6 registers][c0, c1][limb]")
    o.append("BN254_CONSTANT uint32_t KCONST[18][2][9] = {\n    " + ",\n    ".join("{%s, %s}" % (mont_limbs(c[0]), mont_limbs(c[1])) for P in (1, 2, 3) for c in frob_constants(P)) + "};")
    mc = sorted(miller_constants().items())
    o.append("// constants of the Miller program (isomorphic-curve scalings, +-(27 - 3i), twist Frobenius coefficients): register offset, limbs")
    o.append("constexpr int NMCONST = %d;" % len(mc))
    o.append("BN254_CONSTANT uint32_t MCONST_OFF[%d] = {%s};" % (len(mc), ", ".join(str(off(r)) for r, _ in mc)))
    o.append("BN254_CONSTANT uint32_t MCONST[%d][2][9] = {\n    %s};" % (len(mc), ",\n    ".join("{%s, %s}" % (mont_limbs(c[0]), mont_limbs(c[1])) for _, c in mc)))
    o.append("constexpr int OFF_IN_P = %d, OFF_IN_Q = %d, OFF_C_ONE = %d;      // inputs of the Miller program: P as (x, 0), (y, 0), (z, 0); Q; the constant 1" % (off(IN_PX), off(IN_QX), off(C_ONE)))
    assert [IN_PY, IN_PZ] == [IN_PX + 1, IN_PX + 2] and [IN_QY, IN_QZ] == [IN_QX + 1, IN_QX + 2] and (IN_PX >> 5) == (IN_PZ >> 5) and (IN_QX >> 5) == (IN_QZ >> 5)
    o.append("constexpr int NPHASES = %d, NPHASES_FE = %d, MULR_PHASE0 = %d;      // tables 0 .. NPHASES_FE-1 serve the programs without a Miller loop, MULR_PHASE0 .. +2 the product of the tree" % (len(B.phases), B.nfe, B.mulr0))
    o.append("// one role per lane pair and phase: 10 source indices (BYTE offsets of even slots, limb 0; REL = relative to the entry's base), dst, flags")
    o.append("struct Role { uint16_t src[10]; uint16_t dst; uint16_t flags; };")
    o.append("BN254_CONSTANT Role ROLES[NPHASES][32] = {")
    for ph in B.phases:
        rows = []
        for pi in range(32):
            w = role_words(ph.kind, ph.roles[pi] if pi < len(ph.roles) else None)
            rows.append("{{%s}, %d, %d}" % (", ".join(str(x) for x in w[:10]), w[10], w[11]))
        o.append("    /* %-10s %-9s */ {%s}," % (ph.name, ph.kind, ", ".join(rows)))
    o.append("};")
    o.append("// program entry: op | phase << 4 | base offset << 12")
    for name, prog in progs.items():
        words = [op | (pid << 4) | (b << 12) for (op, pid, b) in prog]
        o.append("constexpr int PROG_%s_LEN = %d;" % (name, len(words)))
        o.append("BN254_CONSTANT uint32_t PROG_%s[%d] = {%s};" % (name, len(words), ", ".join("0x%xu" % w for w in words)))
    o.append("}}  // namespace bn254::wv")
    text = "\n".join(o) + "\n"
    p = pathlib.Path(path)
    if not p.exists() or p.read_text() != text:
        p.write_text(text); print("wrote", p)
    else:
        print("up to date:", p)

if __name__ == "__main__":
    B, progs = build()
    self_check(B, progs)
    counts = {}
    for (op, pid, b) in progs["FE"]: counts[op] = counts.get(op, 0) + 1
    print("FE program: %d phases" % (len(progs["FE"]) - 1), {k: counts.get(v, 0) for k, v in OPS.items()}, "Miller program: %d phases" % (len(progs["MILLER"]) - 1), "tables:", len(B.phases))
    emit(B, progs, sys.argv[1] if len(sys.argv) > 1 else pathlib.Path(__file__).resolve().parents[1] / "bn_amd" / "csrc" / "wave_tables.hpp")

"""TEST INFRASTRUCTURE - big-integer model of the reference `bn` crate's pairing path.

This is a checker, not product code: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  It is the *value-level* restatement
(plain integers mod q, no Montgomery limbs): every constant is derived from the BN
parameter u and every formula follows the reference, so results are the canonical
values the reference would print.  The limb-level, reference-faithful restatement
(SOS Montgomery with 32-bit halves, binary-EEA inversion) is oracle/bn_oracle.c;
the two are cross-checked against each other and both are pinned by the reference's
own known-answer tests (tests/golden/reference_kats.json, tests/test_oracle_kats.py).

Reference citations (relative to /root/reference):
  Fq2   src/fields/fq2.rs:63-154      Fq6  src/fields/fq6.rs:59-157
  Fq12  src/fields/fq12.rs:41-306     G    src/groups/mod.rs:113-347
  pairing pieces  src/groups/mod.rs:441-635, 764-771
"""

U = 4965661367192848881                      # BN parameter (fq12.rs:98-100)
Q = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1     # base field modulus (fp.rs:172)
R_ORD = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1  # group order (fp.rs:163)
ATE_LOOP_COUNT = 6 * U + 2                   # groups/mod.rs:452-454
MONT_R = 1 << 256                            # Montgomery radix of the 4x64 limb image

assert Q == 21888242871839275222246405745257275088696311157297823662689037894645226208583
assert R_ORD == 21888242871839275222246405745257275088548364400416034343698204186575808495617
assert ATE_LOOP_COUNT == 0x19d797039be763ba8


def inv(a, m=Q):
    return pow(a, -1, m)

# ---------------------------------------------------------------- Fq2 = Fq[i]/(i^2+1)
def f2(a, b=0): return (a % Q, b % Q)
F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (9, 1)                                   # fq2_nonresidue (fq2.rs:17-22), twist (groups/mod.rs:442)

def f2_add(x, y): return ((x[0] + y[0]) % Q, (x[1] + y[1]) % Q)
def f2_sub(x, y): return ((x[0] - y[0]) % Q, (x[1] - y[1]) % Q)
def f2_neg(x): return ((-x[0]) % Q, (-x[1]) % Q)
def f2_mul(x, y):
    return ((x[0] * y[0] - x[1] * y[1]) % Q, (x[0] * y[1] + x[1] * y[0]) % Q)
def f2_sqr(x): return f2_mul(x, x)
def f2_scale(x, k): return (x[0] * k % Q, x[1] * k % Q)
def f2_mul_xi(x): return f2_mul(x, XI)
def f2_conj(x): return (x[0], (-x[1]) % Q)
def f2_frob(x, p): return x if p % 2 == 0 else f2_conj(x)   # fq2.rs:74-83
def f2_inv(x):                                              # fq2.rs:125-136
    t = inv((x[0] * x[0] + x[1] * x[1]) % Q)
    return (x[0] * t % Q, (-x[1] * t) % Q)
def f2_pow(x, e):
    r = F2_ONE
    for bit in bin(e)[2:]:
        r = f2_sqr(r)
        if bit == '1':
            r = f2_mul(r, x)
    return r

# Frobenius / twist constants, all derived: gamma_{k}(p) = xi^((q^p - 1) * k / 6)
def _gamma(num, den, p): return f2_pow(XI, (Q**p - 1) * num // den)
FROB6_C1 = [_gamma(1, 3, p) for p in range(4)]      # fq6.rs:5-22
FROB6_C2 = [_gamma(2, 3, p) for p in range(4)]      # fq6.rs:23-40
FROB12_C1 = [_gamma(1, 6, p) for p in range(4)]     # fq12.rs:7-24
TWIST_MUL_BY_Q_X = _gamma(1, 3, 1)                  # groups/mod.rs:456-461
TWIST_MUL_BY_Q_Y = _gamma(1, 2, 1)                  # groups/mod.rs:464-469
TWO_INV = inv(2)                                    # groups/mod.rs:446-449
G1_B = 3                                            # groups/mod.rs:363-365
G2_B = f2_mul((3, 0), f2_inv(XI))                   # groups/mod.rs:392-397  (b' = 3/xi)
G1_ONE = (1, 2, 1)                                  # groups/mod.rs:355-361
G2_ONE = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531),
          F2_ONE)                                   # standard alt_bn128 G2 generator (groups/mod.rs:377-390)

# ---------------------------------------------------------------- Fq6 = Fq2[v]/(v^3 - xi)
F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)
def f6_add(x, y): return tuple(f2_add(a, b) for a, b in zip(x, y))
def f6_sub(x, y): return tuple(f2_sub(a, b) for a, b in zip(x, y))
def f6_neg(x): return tuple(f2_neg(a) for a in x)
def f6_mul_v(x): return (f2_mul_xi(x[2]), x[0], x[1])          # fq6.rs:59-65
def f6_scale(x, k): return tuple(f2_mul(a, k) for a in x)
def f6_mul(x, y):                                              # fq6.rs:147-157
    aa, bb, cc = f2_mul(x[0], y[0]), f2_mul(x[1], y[1]), f2_mul(x[2], y[2])
    c0 = f2_add(f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(x[1], x[2]), f2_add(y[1], y[2])), bb), cc)), aa)
    c1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(x[0], x[1]), f2_add(y[0], y[1])), aa), bb), f2_mul_xi(cc))
    c2 = f2_sub(f2_add(f2_sub(f2_mul(f2_add(x[0], x[2]), f2_add(y[0], y[2])), aa), bb), cc)
    return (c0, c1, c2)
def f6_sqr(x): return f6_mul(x, x)
def f6_frob(x, p):                                             # fq6.rs:75-81
    return (f2_frob(x[0], p), f2_mul(f2_frob(x[1], p), FROB6_C1[p]), f2_mul(f2_frob(x[2], p), FROB6_C2[p]))
def f6_inv(x):                                                 # fq6.rs:129-141
    c0 = f2_sub(f2_sqr(x[0]), f2_mul(x[1], f2_mul_xi(x[2])))
    c1 = f2_sub(f2_mul_xi(f2_sqr(x[2])), f2_mul(x[0], x[1]))
    c2 = f2_sub(f2_sqr(x[1]), f2_mul(x[0], x[2]))
    t = f2_inv(f2_add(f2_mul_xi(f2_add(f2_mul(x[2], c1), f2_mul(x[1], c2))), f2_mul(x[0], c0)))
    return (f2_mul(t, c0), f2_mul(t, c1), f2_mul(t, c2))

# ---------------------------------------------------------------- Fq12 = Fq6[w]/(w^2 - v)
F12_ONE = (F6_ONE, F6_ZERO)
def f12_add(x, y): return (f6_add(x[0], y[0]), f6_add(x[1], y[1]))
def f12_sub(x, y): return (f6_sub(x[0], y[0]), f6_sub(x[1], y[1]))
def f12_neg(x): return (f6_neg(x[0]), f6_neg(x[1]))
def f12_mul(x, y):                                             # fq12.rs:298-306
    aa, bb = f6_mul(x[0], y[0]), f6_mul(x[1], y[1])
    return (f6_add(f6_mul_v(bb), aa),
            f6_sub(f6_sub(f6_mul(f6_add(x[0], x[1]), f6_add(y[0], y[1])), aa), bb))
def f12_sqr(x): return f12_mul(x, x)
def f12_conj(x): return (x[0], f6_neg(x[1]))                   # unitary_inverse fq12.rs:103-105
def f12_inv(x):                                                # fq12.rs:284-292
    t = f6_inv(f6_sub(f6_sqr(x[0]), f6_mul_v(f6_sqr(x[1]))))
    return (f6_mul(x[0], t), f6_neg(f6_mul(x[1], t)))
def f12_frob(x, p):                                            # fq12.rs:90-95
    return (f6_frob(x[0], p), f6_scale(f6_frob(x[1], p), FROB12_C1[p]))
def f12_mul_by_024(f, ell_0, ell_vw, ell_vv):                  # fq12.rs:107-176 (value: f * sparse)
    sparse = ((ell_0, F2_ZERO, ell_vv), (F2_ZERO, ell_vw, F2_ZERO))   # x0 + x2 v^2 + x4 v w
    return f12_mul(f, sparse)
def f12_cyclotomic_squared(x):                                 # fq12.rs:178-227 (Granger-Scott; NOT x*x off the subgroup)
    z0, z4, z3 = x[0]; z2, z1, z5 = x[1]
    def fp4_sq(a, b):                                          # (a + b s)^2 with s^2 = xi -> (a^2 + xi b^2, 2ab)
        t = f2_mul(a, b)
        return (f2_sub(f2_sub(f2_mul(f2_add(a, b), f2_add(f2_mul_xi(b), a)), t), f2_mul_xi(t)), f2_add(t, t))
    t0, t1 = fp4_sq(z0, z1); t2, t3 = fp4_sq(z2, z3); t4, t5 = fp4_sq(z4, z5)
    def m3s(t, z):                                             # 2(t - z) + t
        d = f2_sub(t, z); return f2_add(f2_add(d, d), t)
    def m3a(t, z):                                             # 2(t + z) + t
        d = f2_add(t, z); return f2_add(f2_add(d, d), t)
    z0 = m3s(t0, z0); z1 = m3a(t1, z1)
    z2 = m3a(f2_mul_xi(t5), z2); z3 = m3s(t4, z3)
    z4 = m3s(t2, z4); z5 = m3a(t3, z5)
    return ((z0, z4, z3), (z2, z1, z5))
def f12_cyclotomic_pow(x, e):                                  # fq12.rs:229-246
    r = F12_ONE; found = False
    for bit in bin(e)[2:]:
        if found:
            r = f12_cyclotomic_squared(r)
        if bit == '1':
            found = True
            r = f12_mul(x, r)
    return r
def f12_exp_by_neg_z(x): return f12_conj(f12_cyclotomic_pow(x, U))    # fq12.rs:97-101
def f12_pow(x, e):                                                    # fields/mod.rs:35-46 (e canonical integer)
    r = F12_ONE
    for i in range(255, -1, -1):
        r = f12_sqr(r)
        if (e >> i) & 1:
            r = f12_mul(x, r)
    return r

def final_exp_first_chunk(f):                                  # fq12.rs:41-52
    c = f12_mul(f12_conj(f), f12_inv(f))
    return f12_mul(f12_frob(c, 2), c)
def final_exp_last_chunk(s):                                   # fq12.rs:54-84
    a = f12_exp_by_neg_z(s); b = f12_cyclotomic_squared(a); c = f12_cyclotomic_squared(b); d = f12_mul(c, b)
    e = f12_exp_by_neg_z(d); f = f12_cyclotomic_squared(e); g = f12_exp_by_neg_z(f)
    h = f12_conj(d); i = f12_conj(g)
    j = f12_mul(i, e); k = f12_mul(j, h); l = f12_mul(k, b); m = f12_mul(k, e); n = f12_mul(s, m)
    o = f12_frob(l, 1); p = f12_mul(o, n)
    q = f12_frob(k, 2); r = f12_mul(q, p)
    ss = f12_conj(s); t = f12_mul(ss, l); u = f12_frob(t, 3)
    return f12_mul(u, r)
def final_exponentiation(f): return final_exp_last_chunk(final_exp_first_chunk(f))

# ---------------------------------------------------------------- curve groups (Jacobian), generic over field ops
class _Ops:
    def __init__(s, add, sub, mul, neg, inv_, zero, one):
        s.add, s.sub, s.mul, s.neg, s.inv, s.zero, s.one = add, sub, mul, neg, inv_, zero, one
FQ_OPS = _Ops(lambda a, b: (a + b) % Q, lambda a, b: (a - b) % Q, lambda a, b: a * b % Q,
              lambda a: (-a) % Q, inv, 0, 1)
FQ2_OPS = _Ops(f2_add, f2_sub, f2_mul, f2_neg, f2_inv, F2_ZERO, F2_ONE)

def g_zero(o): return (o.zero, o.one, o.zero)                  # groups/mod.rs:208-214
def g_is_zero(o, p): return p[2] == o.zero
def g_double(o, p):                                            # groups/mod.rs:228-247
    x, y, z = p
    a = o.mul(x, x); b = o.mul(y, y); c = o.mul(b, b)
    t = o.add(x, b); d = o.sub(o.sub(o.mul(t, t), a), c); d = o.add(d, d)
    e = o.add(o.add(a, a), a); f = o.mul(e, e)
    x3 = o.sub(f, o.add(d, d))
    c8 = o.add(c, c); c8 = o.add(c8, c8); c8 = o.add(c8, c8)
    yz = o.mul(y, z)
    return (x3, o.sub(o.mul(e, o.sub(d, x3)), c8), o.add(yz, yz))
def g_add(o, p, q):                                            # groups/mod.rs:275-311
    if g_is_zero(o, p): return q
    if g_is_zero(o, q): return p
    z1s = o.mul(p[2], p[2]); z2s = o.mul(q[2], q[2])
    u1 = o.mul(p[0], z2s); u2 = o.mul(q[0], z1s)
    s1 = o.mul(p[1], o.mul(q[2], z2s)); s2 = o.mul(q[1], o.mul(p[2], z1s))
    if u1 == u2 and s1 == s2:
        return g_double(o, p)
    h = o.sub(u2, u1); sd = o.sub(s2, s1)
    hh = o.add(h, h); i = o.mul(hh, hh); j = o.mul(h, i); r = o.add(sd, sd); v = o.mul(u1, i)
    s1j = o.mul(s1, j)
    x3 = o.sub(o.sub(o.mul(r, r), j), o.add(v, v))
    zz = o.add(p[2], q[2])
    return (x3, o.sub(o.mul(r, o.sub(v, x3)), o.add(s1j, s1j)),
            o.mul(o.sub(o.sub(o.mul(zz, zz), z1s), z2s), h))
def g_neg(o, p): return p if g_is_zero(o, p) else (p[0], o.neg(p[1]), p[2])
def g_mul(o, p, k):                                            # groups/mod.rs:250-270 (k canonical integer)
    res = g_zero(o); found = False
    for i in range(255, -1, -1):
        if found: res = g_double(o, res)
        if (k >> i) & 1:
            found = True
            res = g_add(o, res, p)
    return res
def g_to_affine(o, p):                                         # groups/mod.rs:113-130
    if g_is_zero(o, p): return None
    zi = o.inv(p[2]); zi2 = o.mul(zi, zi)
    return (o.mul(p[0], zi2), o.mul(p[1], o.mul(zi2, zi)))
def g_normalize(o, p):                                         # lib.rs:88-95
    a = g_to_affine(o, p)
    return p if a is None else (a[0], a[1], o.one)

# ---------------------------------------------------------------- pairing
def _doubling_step(r):                                         # groups/mod.rs:612-634
    x, y, z = r
    a = f2_scale(f2_mul(x, y), TWO_INV); b = f2_sqr(y); c = f2_sqr(z)
    d = f2_add(f2_add(c, c), c); e = f2_mul(G2_B, d); f = f2_add(f2_add(e, e), e)
    g = f2_scale(f2_add(b, f), TWO_INV)
    h = f2_sub(f2_sqr(f2_add(y, z)), f2_add(b, c)); i = f2_sub(e, b); j = f2_sqr(x); e2 = f2_sqr(e)
    nr = (f2_mul(a, f2_sub(b, f)), f2_sub(f2_sqr(g), f2_add(f2_add(e2, e2), e2)), f2_mul(b, h))
    return nr, (f2_mul(XI, i), f2_neg(h), f2_add(f2_add(j, j), j))     # (ell_0, ell_vw, ell_vv)
def _addition_step(r, base):                                   # groups/mod.rs:592-610
    x, y, z = r
    d = f2_sub(x, f2_mul(z, base[0])); e = f2_sub(y, f2_mul(z, base[1]))
    f = f2_sqr(d); g = f2_sqr(e); h = f2_mul(d, f); i = f2_mul(x, f)
    j = f2_sub(f2_add(f2_mul(z, g), h), f2_add(i, i))
    nr = (f2_mul(d, j), f2_sub(f2_mul(e, f2_sub(i, j)), f2_mul(h, y)), f2_mul(z, h))
    return nr, (f2_mul(XI, f2_sub(f2_mul(e, base[0]), f2_mul(d, base[1]))), d, f2_neg(e))
def _mul_by_q(a):                                              # groups/mod.rs:550-555
    return (f2_mul(TWIST_MUL_BY_Q_X, f2_frob(a[0], 1)), f2_mul(TWIST_MUL_BY_Q_Y, f2_frob(a[1], 1)))
def _loop_bits():
    return [int(b) for b in bin(ATE_LOOP_COUNT)[3:]]           # skips the top bit (groups/mod.rs:565-569)
def precompute(q_aff):                                         # groups/mod.rs:557-588
    r = (q_aff[0], q_aff[1], F2_ONE); coeffs = []
    for bit in _loop_bits():
        r, c = _doubling_step(r); coeffs.append(c)
        if bit:
            r, c = _addition_step(r, q_aff); coeffs.append(c)
    q1 = _mul_by_q(q_aff); q2 = _mul_by_q(q1); q2 = (q2[0], f2_neg(q2[1]))
    r, c = _addition_step(r, q1); coeffs.append(c)
    r, c = _addition_step(r, q2); coeffs.append(c)
    return coeffs
def miller_loop(coeffs, p_aff):                                # groups/mod.rs:486-519
    f = F12_ONE; idx = 0
    def app(f, c): return f12_mul_by_024(f, c[0], f2_scale(c[1], p_aff[1]), f2_scale(c[2], p_aff[0]))
    for bit in _loop_bits():
        f = app(f12_sqr(f), coeffs[idx]); idx += 1
        if bit:
            f = app(f, coeffs[idx]); idx += 1
    f = app(f, coeffs[idx]); f = app(f, coeffs[idx + 1])
    return f
def pairing(p, q):                                             # groups/mod.rs:764-771
    pa, qa = g_to_affine(FQ_OPS, p), g_to_affine(FQ2_OPS, q)
    if pa is None or qa is None:
        return F12_ONE
    return final_exponentiation(miller_loop(precompute(qa), pa))

# ---------------------------------------------------------------- flattening / Montgomery limb images
def f12_flat(x):
    """12 canonical integers in the reference memory order c0.c0.c0, c0.c0.c1, c0.c1.c0 ... c1.c2.c1."""
    return [c for six in x for two in six for c in two]
def f12_unflat(v):
    v = [int(a) % Q for a in v]
    return (((v[0], v[1]), (v[2], v[3]), (v[4], v[5])), ((v[6], v[7]), (v[8], v[9]), (v[10], v[11])))
def to_mont_limbs(a, mod=Q):
    """canonical integer -> the 4 x u64 little-endian Montgomery limbs the reference stores (fp.rs:62-70)."""
    m = a * MONT_R % mod
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
def from_mont_limbs(l, mod=Q):
    m = sum(int(x) << (64 * i) for i, x in enumerate(l))
    return m * inv(MONT_R, mod) % mod

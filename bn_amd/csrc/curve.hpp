// Jacobian group arithmetic on G1 (over Fq) and G2 (over Fq2) and scalar multiplication by an Fr, for the HIP engine.
// Reference: src/groups/mod.rs - double :228-247 (dbl-2009-l, a = 0), add :275-311 (add-2007-bl with the zero / equal-point
// branches), Mul<Fr> :250-270 (MSB-first double-and-add on the scalar taken OUT of Montgomery form, fields/fp.rs:15-22),
// to_affine :113-130, normalize lib.rs:88-95.
//
// Two products:
//   * scalar_mul_reference_chain: executes exactly the reference's operation sequence, so the (non-canonical) Jacobian
//     coordinates it returns are the very limbs `G::random` / `G * Fr` produce in the reference.  Used to build benchmark
//     inputs with z != 1 on the device.
//   * the caller may normalize (x/z^2, y/z^3, 1) - the parity definition for `g*_mul_batch`.
#pragma once
#include "pairing.hpp"

namespace bn254 {

// true when the predicate holds in ANY lane of the wave (a wave-uniform guard around work that almost no lane needs)
#if defined(BN_HOSTSIM)
BN_FN bool bn_any(bool b) { return b; }
#else
BN_FN bool bn_any(bool b) { return __any(b) != 0; }
#endif

// ---- field adaptors: the same generic point code runs over Fe (G1) and over an Fq2 mapping (G2) -------------------------
struct FqField {
    using T = Fe;
    static BN_FN T mul(const T &a, const T &b) { return fe_mul(a, b); }
    static BN_FN T sqr(const T &a) { return fe_sqr(a); }
    template <int C1, int C2, int C3> static BN_FN T lc3(const T &x, const T &y, const T &z) { return fe_lc3<C1, C2, C3>(x, y, z); }
    template <int C1, int C2, int C3> static BN_FN T lc3w(const T &x, const T &y, const T &z) { return fe_lc3w<C1, C2, C3>(x, y, z); }
    static BN_FN T sum(const T &x, const T &y) { return fe_norm(fe_add(x, y)); }                  // lazy sum, carries propagated
    static BN_FN T sum3(const T &x, const T &y, const T &z) { return fe_norm(fe_add(fe_add(x, y), z)); }
    static BN_FN T zero() { return fe_zero(); }
    static BN_FN T one() { return fe_one(); }
    static BN_FN T select(bool b, const T &x, const T &y) { return fe_select(b, x, y); }
    static BN_FN bool is_zero(const T &a) { return fe_is_zero(a); }
    static BN_FN bool is_zero_std(const T &a) { return fe_is_zero_std(a); }        // normalized, below 4q
    static BN_FN T inverse(const T &a) { return fe_inverse(a); }
};
template <class F2>
struct Fq2Field {
    using T = F2;
    static BN_FN T mul(const T &a, const T &b) { return f2_mul(a, b); }
    static BN_FN T sqr(const T &a) { return f2_sqr(a); }
    template <int C1, int C2, int C3> static BN_FN T lc3(const T &x, const T &y, const T &z) { return f2_lc3<C1, C2, C3>(x, y, z); }
    template <int C1, int C2, int C3> static BN_FN T lc3w(const T &x, const T &y, const T &z) { return f2_lc3w<C1, C2, C3>(x, y, z); }
    static BN_FN T sum(const T &x, const T &y) { return f2_sum_for_mul(x, y); }
    static BN_FN T sum3(const T &x, const T &y, const T &z) { return f2_sum3_for_mul(x, y, z); }
    static BN_FN T zero() { return f2_zero((const F2 *)nullptr); }
    static BN_FN T one() { return f2_one((const F2 *)nullptr); }
    static BN_FN T select(bool b, const T &x, const T &y) { return f2_select(b, x, y); }
    static BN_FN bool is_zero(const T &a) { return f2_is_zero(a); }
    static BN_FN bool is_zero_std(const T &a) { return f2_is_zero_std(a); }
    static BN_FN T inverse(const T &a) { return f2_inverse(a); }
};

template <class F> struct Jac { typename F::T x, y, z; };

// groups/mod.rs:228-247
template <class F>
BN_COARSE Jac<F> jac_double(const Jac<F> &p) {
    using T = typename F::T;
    T a = F::sqr(p.x), b = F::sqr(p.y), c = F::sqr(b);
    T t = F::sqr(F::sum(p.x, b));
    T d = F::template lc3w<2, -2, -2>(t, a, c);
    T e = F::sum3(a, a, a);
    T f = F::sqr(e);
    Jac<F> r;
    r.x = F::template lc3<1, -2, 0>(f, d, d);
    r.y = F::template lc3<1, -8, 0>(F::mul(e, F::template lc3<1, -1, 0>(d, r.x, d)), c, c);
    { T yz = F::mul(p.y, p.z); r.z = F::sum(yz, yz); }
    return r;
}
// The same doubling over Fq (G1) with the two places where a dual product pays (fe_mul2: a u + c v with ONE reduction):
//   D = 2((X + B)^2 - A - C) = 4 X B        one product of the lazy 4X instead of a square, a carry propagation and a wide reduction
//   Y3 = E (D - X3) - 8 B^2 = E (D - X3) + (4B)(-2B)     one dual product: C = B^2 is never formed, nor the two reductions around it
// 3 squares + 2 products + 1 dual product (945 multiply-adds) instead of 5 + 2 (954), and ~240 fewer other instructions per doubling
// (of ~1650).  Same field elements X3, Y3, Z3 as groups/mod.rs:228-247, hence the same bytes wherever they are stored.
BN_COARSE Jac<FqField> jac_double(const Jac<FqField> &p) {
    const Fe a = fe_sqr(p.x), b = fe_sqr(p.y);
    const Fe x2 = fe_add(p.x, p.x);
    const Fe d = fe_mul(fe_add(x2, x2), b);                                      // 4 X B  (limbs of 4X stay below 2^31)
    const Fe e = fe_norm(fe_add(fe_add(a, a), a));
    const Fe f = fe_sqr(e);
    Jac<FqField> r;
    r.x = fe_lc3<1, -2, 0>(f, d, d);
    const Fe nb = fe_neg<1, 3>(b);                                               // -B, lazy
    r.y = fe_mul2(e, fe_sub<1, 3>(d, r.x), fe_norm(fe_add(fe_add(b, b), fe_add(b, b))), fe_norm(fe_add(nb, nb)));
    { const Fe yz = fe_mul(p.y, p.z); r.z = fe_norm(fe_add(yz, yz)); }
    return r;
}
// out-of-line copy for the never-taken equal-points branch below (keeps the inlined builds small).  On the GPU the operands travel
// BY VALUE as <9 x i32> vectors (fe.hpp, "leaf calling convention"): a Jac handed over by reference has to live in private memory,
// and the compiler then stores the running point there after EVERY addition of a scalar-multiplication chain, branch taken or not
// - 27 dwords x 99 steps = 10.7 KB of HBM writes per G1 multiplication (round 3: profiles/r03z_pmc_side.txt, WRITE_SIZE 2.5 GB per
// 2^18).  Only the result comes back through memory, inside the cold block.
template <class F>
BN_OUTER Jac<F> jac_double_cold_ref(const Jac<F> &p) { return jac_double(p); }
#if defined(BN_HOSTSIM)
template <class F>
BN_FN Jac<F> jac_double_cold(const Jac<F> &p) { return jac_double_cold_ref(p); }
#else
BN_FN u32x9 jac_coord_vec(const Fe &a) { return bn_tov(a); }
template <class T> BN_FN u32x9 jac_coord_vec(const Fq2B<T> &a) { return bn_tov(a.v); }
BN_FN u32x9 jac_coord_vec(const Fq2A &a) { return bn_tov(a.c0); }                 // never used: two-Fe coordinates take the by-reference path
BN_FN void jac_coord_set(Fe &a, u32x9 v) { a = bn_unv(v); }
template <class T> BN_FN void jac_coord_set(Fq2B<T> &a, u32x9 v) { a.v = bn_unv(v); }
BN_FN void jac_coord_set(Fq2A &, u32x9) {}
template <class F>
BN_OUTER void jac_double_cold_vec(u32x9 x, u32x9 y, u32x9 z, Jac<F> *out) {
    Jac<F> p;
    jac_coord_set(p.x, x); jac_coord_set(p.y, y); jac_coord_set(p.z, z);
    *out = jac_double(p);
}
template <class F>
BN_FN Jac<F> jac_double_cold(const Jac<F> &p) {
    if constexpr (sizeof(typename F::T) == sizeof(Fe)) {
        Jac<F> d;
        jac_double_cold_vec<F>(jac_coord_vec(p.x), jac_coord_vec(p.y), jac_coord_vec(p.z), &d);
        return d;
    } else {
        return jac_double_cold_ref(p);
    }
}
#endif
// groups/mod.rs:275-311, all branches (zero operands, equal points) as per-lane selects; the doubling for equal points is a
// divergent branch that no lane takes for valid prime-order inputs and scalars < r.  pz / qz: "p (q) is the point at infinity".
template <class F>
BN_COARSE Jac<F> jac_add_flags(const Jac<F> &p, const Jac<F> &q, bool pz, bool qz) {
    using T = typename F::T;
    // ordered for short live ranges: the z-dependent values first, Z3 as soon as H exists (the operands stay live to the end only
    // for the selects of the special cases below)
    T z1s = F::sqr(p.z);
    T u2 = F::mul(q.x, z1s), s2 = F::mul(q.y, F::mul(p.z, z1s));
    T z2s = F::sqr(q.z);
    T u1 = F::mul(p.x, z2s), s1 = F::mul(p.y, F::mul(q.z, z2s));
    T h = F::template lc3<1, -1, 0>(u2, u1, u1), sd = F::template lc3<1, -1, 0>(s2, s1, s1);
    bool same = F::is_zero_std(h) && F::is_zero_std(sd) && !pz && !qz;          // h, sd are fused reductions: normalized, < 2q
    Jac<F> r;
    r.z = F::mul(F::template lc3<1, -1, -1>(F::sqr(F::template lc3<1, 1, 0>(p.z, q.z, q.z)), z1s, z2s), h);
    T i = F::sqr(F::sum(h, h));
    T j = F::mul(h, i);
    T v = F::mul(u1, i);
    T rr = F::sum(sd, sd);
    r.x = F::template lc3<1, -1, -2>(F::sqr(rr), j, v);
    r.y = F::template lc3<1, -2, 0>(F::mul(rr, F::template lc3<1, -1, 0>(v, r.x, v)), F::mul(s1, j), j);
    if (same) {                                    // groups/mod.rs:291-292
        const Jac<F> pc = p;                       // a copy: `p` itself must not escape by reference (it would live in memory)
        Jac<F> d = jac_double_cold(pc);
        r.x = F::select(same, r.x, d.x); r.y = F::select(same, r.y, d.y); r.z = F::select(same, r.z, d.z);
    }
    r.x = F::select(qz, r.x, p.x); r.y = F::select(qz, r.y, p.y); r.z = F::select(qz, r.z, p.z);     // :280-282
    r.x = F::select(pz, r.x, q.x); r.y = F::select(pz, r.y, q.y); r.z = F::select(pz, r.z, q.z);     // :276-278
    return r;
}
// Mixed addition p + (q.x, q.y, 1): the same add-2007-bl with Z2 = 1 (8 products + 3 squarings instead of 11 + 5), same special
// cases.  Used with window tables that were brought to a COMMON z and are therefore affine points of an isomorphic curve (below).
template <class F> struct Aff { typename F::T x, y; };
template <class F>
BN_COARSE Jac<F> jac_madd_flags(const Jac<F> &p, const Aff<F> &q, bool pz, bool qz) {
    using T = typename F::T;
    T z1s = F::sqr(p.z);
    T u2 = F::mul(q.x, z1s), s2 = F::mul(q.y, F::mul(p.z, z1s));
    T h = F::template lc3<1, -1, 0>(u2, p.x, p.x), sd = F::template lc3<1, -1, 0>(s2, p.y, p.y);
    bool same = F::is_zero_std(h) && F::is_zero_std(sd) && !pz && !qz;
    Jac<F> r;
    { T zh = F::mul(p.z, h); r.z = F::sum(zh, zh); }
    T i = F::sqr(F::sum(h, h));
    T j = F::mul(h, i);
    T v = F::mul(p.x, i);
    T rr = F::sum(sd, sd);
    r.x = F::template lc3<1, -1, -2>(F::sqr(rr), j, v);
    r.y = F::template lc3<1, -2, 0>(F::mul(rr, F::template lc3<1, -1, 0>(v, r.x, v)), F::mul(p.y, j), j);
    if (same) {
        const Jac<F> pc = p;
        Jac<F> d = jac_double_cold(pc);
        r.x = F::select(same, r.x, d.x); r.y = F::select(same, r.y, d.y); r.z = F::select(same, r.z, d.z);
    }
    r.x = F::select(pz, r.x, q.x); r.y = F::select(pz, r.y, q.y); r.z = F::select(pz, r.z, F::one());     // infinity + q = q ...
    r.x = F::select(qz, r.x, p.x); r.y = F::select(qz, r.y, p.y); r.z = F::select(qz, r.z, p.z);           // ... and p + infinity = p, also when p is infinite
    return r;
}
// The same mixed addition over Fq (G1) with  Y3 = r (V - X3) - 2 Y1 J  as ONE dual product (fe_mul2) instead of two products and two
// fused reductions: 81 multiply-adds and ~115 other instructions fewer per addition.  Same field elements, same special cases.
BN_COARSE Jac<FqField> jac_madd_flags(const Jac<FqField> &p, const Aff<FqField> &q, bool pz, bool qz) {
    using F = FqField;
    const Fe z1s = fe_sqr(p.z);
    const Fe u2 = fe_mul(q.x, z1s), s2 = fe_mul(q.y, fe_mul(p.z, z1s));
    const Fe h = fe_lc3<1, -1, 0>(u2, p.x, p.x), sd = fe_lc3<1, -1, 0>(s2, p.y, p.y);
    const bool same = fe_is_zero_std(h) && fe_is_zero_std(sd) && !pz && !qz;
    Jac<F> r;
    { const Fe zh = fe_mul(p.z, h); r.z = fe_norm(fe_add(zh, zh)); }
    const Fe i = fe_sqr(fe_norm(fe_add(h, h)));
    const Fe j = fe_mul(h, i);
    const Fe v = fe_mul(p.x, i);
    const Fe rr = fe_norm(fe_add(sd, sd));
    r.x = fe_lc3<1, -1, -2>(fe_sqr(rr), j, v);
    const Fe ny = fe_neg<1, 4>(p.y);                                             // -Y1, lazy (Y1 < 3q)
    r.y = fe_mul2(rr, fe_sub<1, 3>(v, r.x), fe_norm(fe_add(ny, ny)), j);
    if (same) {
        const Jac<F> pc = p;
        Jac<F> d = jac_double_cold(pc);
        r.x = F::select(same, r.x, d.x); r.y = F::select(same, r.y, d.y); r.z = F::select(same, r.z, d.z);
    }
    r.x = F::select(pz, r.x, q.x); r.y = F::select(pz, r.y, q.y); r.z = F::select(pz, r.z, F::one());
    r.x = F::select(qz, r.x, p.x); r.y = F::select(qz, r.y, p.y); r.z = F::select(qz, r.z, p.z);
    return r;
}
// The mixed addition of the G1 window loop (scalar_mul_glv) with its bookkeeping fused in - same field elements X3, Y3, Z3 as above for
// p + (q.x, +-q.y):
//   * the table entry's sign (Booth digit, GLV half) is applied where y FIRST meets the accumulator: sd = +-S2 - Y1 is one fused reduction
//     with a per-lane coefficient (fe_lc3_core SIGN2) instead of a full negation of q.y (a reduction of its own) and a select;
//   * "the accumulator is at infinity afterwards" is read off H: Z3 = 2 Z1 H vanishes exactly when H does (Z1 != 0), which was tested anyway
//     for the equal-points branch - instead of testing Z3 again in the caller;
//   * S2 - Y1 == 0 is only evaluated when H == 0 in some lane of the wave (never, for valid inputs), and the signed y that an accumulator
//     at infinity adopts only when some lane's accumulator is at infinity (the first one or two windows).
// `pz`: the accumulator is at infinity; `qz`: the operand is (digit 0); returns the sum and updates `pz`.
BN_COARSE Jac<FqField> jac_madd_signed(const Jac<FqField> &p, const Aff<FqField> &q, bool negate, bool &pz, bool qz) {
    using F = FqField;
    const Fe z1s = fe_sqr(p.z);
    const Fe u2 = fe_mul(q.x, z1s), s2 = fe_mul(q.y, fe_mul(p.z, z1s));
    const Fe h = fe_lc3<1, -1, 0>(u2, p.x, p.x), sd = fe_lc3_core<-1, 1, 0, true>(p.y, s2, s2, negate);
    const bool hz = fe_is_zero_std(h);
    bool same = false;
    if (bn_any(hz)) same = hz && fe_is_zero_std(sd) && !pz && !qz;
    Jac<F> r;
    { const Fe zh = fe_mul(p.z, h); r.z = fe_norm(fe_add(zh, zh)); }
    const Fe i = fe_sqr(fe_norm(fe_add(h, h)));
    const Fe j = fe_mul(h, i);
    const Fe v = fe_mul(p.x, i);
    const Fe rr = fe_norm(fe_add(sd, sd));
    r.x = fe_lc3<1, -1, -2>(fe_sqr(rr), j, v);
    const Fe ny = fe_neg<1, 4>(p.y);                                             // -Y1, lazy (Y1 < 3q)
    r.y = fe_mul2(rr, fe_sub<1, 3>(v, r.x), fe_norm(fe_add(ny, ny)), j);
    if (same) {
        const Jac<F> pc = p;
        Jac<F> d = jac_double_cold(pc);
        r.x = F::select(same, r.x, d.x); r.y = F::select(same, r.y, d.y); r.z = F::select(same, r.z, d.z);
    }
    if (bn_any(pz)) {                                                            // infinity + q = q, with q's sign
        const Fe qy = F::select(negate, q.y, fe_lc3<-1, 0, 0>(q.y, q.y, q.y));
        r.x = F::select(pz, r.x, q.x); r.y = F::select(pz, r.y, qy); r.z = F::select(pz, r.z, F::one());
    }
    r.x = F::select(qz, r.x, p.x); r.y = F::select(qz, r.y, p.y); r.z = F::select(qz, r.z, p.z);     // p + infinity = p, also when p is infinite
    pz = qz ? pz : (!pz && hz && !same);                                         // opposite points (H == 0, S2 != Y1) sum to infinity
    return r;
}
// Window table -> COMMON z without an inversion.  Entry i = (X_i : Y_i : Z_i) is rescaled by s_i = prod_{j != i} Z_j to
// (X_i s_i^2 : Y_i s_i^3 : Zc), Zc = prod Z_j.  On the isomorphic curve y^2 = x^3 + b Zc^6 - reached by (x, y) -> (x Zc^2, y Zc^3), and
// the group law of a curve with a = 0 never looks at b - the rescaled (X, Y) are AFFINE points: the whole chain runs there with
// mixed additions and the result (X : Y : Z) is the point (X : Y : Z Zc) of the original curve.  Cost: 3N - 4 + 4N products for N
// entries (52 for N = 8) against 5 saved per addition (66 additions: 330), and a table entry is 2 instead of 3 field elements.
// (The "effective affine" technique of libsecp256k1's ecmult, with the plain prefix/suffix product instead of its z-ratio chain.)
// tab[1..N] in, aff[1..N] out; returns Zc.  An infinite input point gives Zc = 0 and the chain's result z = 0: infinity again.
// CONJ_EXTRA (G2 only): every entry is rescaled by conj(Zc) as well, so that the common z becomes Zc conj(Zc) = norm(Zc), an element
// of Fq - the isomorphism (x, y) -> (x s^2, y s^3) commutes with the Frobenius-twist endomorphism psi exactly when s is in Fq.
// Where the affine window table lives.  Default: a per-lane array (host simulation, one-lane mapping).  The kernels keep it in a
// buffer laid out [lane][entry][18 dwords padded to 80 bytes] (bn254_kernels_mul.hip AffTableMem): the entry a lane reads depends
// on ITS digit, and private (scratch) memory is interleaved across lanes dword by dword - a lane-indexed private array makes every
// dword load of a wave touch up to 64 different 256-byte rows (measured: 32-51 GB of traffic per 2^20 G1 multiplications, 11 % of
// the kernel's time, profiles/r03m_pmc_side.txt); 80 contiguous bytes per lane cost two cache lines.
template <class F>
struct AffTableVars {
    Aff<F> e[9];
    BN_FN void put(int i, const Aff<F> &v) { e[i] = v; }
    BN_FN Aff<F> get(int i) const { return e[i]; }
};
template <class F, int N, bool CONJ_EXTRA = false, class Tab>
BN_FN typename F::T table_to_common_z(const Jac<F> *tab, Tab &aff) {
    using T = typename F::T;
    T pre[N + 1];                                             // pre[i] = Z_1 ... Z_i
    pre[1] = tab[1].z;
#pragma unroll 1
    for (int i = 2; i <= N; ++i) pre[i] = F::mul(pre[i - 1], tab[i].z);
    T suf = F::one();                                         // (conj(Zc)) Z_{i+1} ... Z_N
    if constexpr (CONJ_EXTRA) suf = f2_conj(pre[N]);
#pragma unroll 1
    for (int i = N; i >= 1; --i) {
        T s = i > 1 ? F::mul(pre[i - 1], suf) : suf;
        T s2 = F::sqr(s);
        aff.put(i, Aff<F>{F::mul(tab[i].x, s2), F::mul(tab[i].y, F::mul(s2, s))});
        if (i > 1) suf = F::mul(suf, tab[i].z);
    }
    return pre[N];
}
template <class F>
BN_FN Jac<F> jac_add(const Jac<F> &p, const Jac<F> &q) { return jac_add_flags(p, q, F::is_zero(p.z), F::is_zero(q.z)); }

// Fr out of Montgomery form (fields/fp.rs:15-22: multiply by 1): 8 x u32 words, word-serial Montgomery reduction mod r
BN_FN void fr_from_mont(const uint32_t *km, uint32_t *raw) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = km[i];
    t[8] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t m = t[0] * k::FR_INV32;
        uint64_t c = ((uint64_t)m * k::FR_MOD32[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            uint64_t x = (uint64_t)m * k::FR_MOD32[j] + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        uint64_t x = (uint64_t)t[8] + c;
        t[7] = (uint32_t)x;
        t[8] = (uint32_t)(x >> 32);
    }
    // t < 2r; one conditional subtraction
    uint32_t d[8];
    int64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int64_t s = (int64_t)t[i] - (int64_t)k::FR_MOD32[i] + br;
        d[i] = (uint32_t)s;
        br = s >> 32;
    }
    bool ge = (t[8] != 0) || (br == 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) raw[i] = ge ? d[i] : t[i];
}

// groups/mod.rs:250-270: res = 0; for bits MSB->LSB { if found { res = 2 res }; if bit { found = true; res = res + p } }.
// Per-lane scalars differ, so `found`/`bit` are per-lane predicates applied by selects; the op sequence each lane's result
// went through is exactly the reference's.
template <class F>
BN_FN Jac<F> scalar_mul_reference_chain(const Jac<F> &p, const uint32_t *k_raw) {
    Jac<F> res = {F::zero(), F::one(), F::zero()};           // groups/mod.rs:208-214
    bool found = false;
#pragma unroll 1
    for (int i = 255; i >= 0; --i) {
        Jac<F> d = jac_double(res);
        res.x = F::select(found, res.x, d.x); res.y = F::select(found, res.y, d.y); res.z = F::select(found, res.z, d.z);
        bool bit = (k_raw[i >> 5] >> (i & 31)) & 1;
        Jac<F> s = jac_add(res, p);
        res.x = F::select(bit, res.x, s.x); res.y = F::select(bit, res.y, s.y); res.z = F::select(bit, res.z, s.z);
        found = found || bit;
    }
    return res;
}
// Fixed 4-bit windows, MSB first: 252 doublings + 64 additions + a 14-operation table instead of the reference chain's
// 256 x (double, add) under per-lane predicates.  Same group element as groups/mod.rs:250-270, different Jacobian
// coordinates - callers normalize (the parity definition of g*_mul_batch).  The 16-entry table is a per-lane array indexed
// by the lane's own digit, i.e. it lives in private memory; infinity is tracked by flags, not by testing z.
template <class F>
BN_FN Jac<F> scalar_mul_windowed(const Jac<F> &p, const uint32_t *k_raw) {
    Jac<F> tab[16];
    const bool p_inf = F::is_zero(p.z);
    tab[0] = {F::zero(), F::one(), F::zero()};
    tab[1] = p;
#pragma unroll 1
    for (int i = 2; i < 16; i += 2) {
        tab[i] = jac_double(tab[i >> 1]);
        tab[i + 1] = jac_add_flags(tab[i], p, p_inf, p_inf);
    }
    Jac<F> res = tab[0];
    bool res_inf = true;
#pragma unroll 1
    for (int w = 63; w >= 0; --w) {
#pragma unroll 1
        for (int d = 0; d < 4; ++d) res = jac_double(res);          // 2^4 * infinity stays infinity (z = 0)
        const uint32_t digit = (k_raw[w >> 3] >> ((w & 7) * 4)) & 15u;
        const bool q_inf = p_inf || digit == 0;
        res = jac_add_flags(res, tab[digit], res_inf, q_inf);
        res_inf = res_inf && q_inf;
    }
    return res;
}
// 4-bit SIGNED (Booth) windows over the affine table 1P .. 8P on the isomorphic curve (table_to_common_z): 252 doublings + 64 mixed
// additions + a table of 4 doublings, 3 additions and 52 products - against scalar_mul_windowed's 64 full additions and 14-operation
// table of 16 Jacobian entries.  A table entry is 2 field elements and there are 8 of them: a third of the private memory.
template <class F, class Tab>
BN_FN Jac<F> scalar_mul_booth_affine(const Jac<F> &p, const uint32_t *k_raw, Tab &aff) {
    const bool p_inf = F::is_zero(p.z);
    Jac<F> tab[9];
    tab[1] = p;
    tab[2] = jac_double(tab[1]);
    tab[4] = jac_double(tab[2]);
    tab[8] = jac_double(tab[4]);
    tab[3] = jac_add_flags(tab[2], p, p_inf, p_inf);
    tab[6] = jac_double(tab[3]);
    tab[5] = jac_add_flags(tab[4], p, p_inf, p_inf);
    tab[7] = jac_add_flags(tab[6], p, p_inf, p_inf);
    const typename F::T zc = table_to_common_z<F, 8>(tab, aff);
    Jac<F> res = {F::zero(), F::one(), F::zero()};
    bool res_inf = true;
#pragma unroll 1
    for (int w = 63; w >= 0; --w) {
        if (w != 63) {
#pragma unroll 1
            for (int d = 0; d < 4; ++d) res = jac_double(res);
        }
        const int d = booth_digit_256(k_raw, w);               // k < r < 2^254: the top window needs no carry
        const int ad = d < 0 ? -d : d;
        Aff<F> q = aff.get(ad ? ad : 1);                        // digit 0: the operand is ignored (q_inf)
        q.y = F::select(d < 0, q.y, F::template lc3<-1, 0, 0>(q.y, q.y, q.y));
        const bool q_inf = p_inf || ad == 0;
        res = jac_madd_flags(res, q, res_inf, q_inf);
        res_inf = res_inf && q_inf;                             // k < r: a partial sum j P with 0 < j < r is never infinity
    }
    res.z = F::mul(res.z, zc);
    return res;
}
// ---- G1 only: GLV.  phi(x, y) = (beta x, y) is multiplication by lambda on the order-r subgroup of E(Fq) (beta^3 = 1, lambda^3 = 1),
// so k P = k1 P + k2 phi(P) with k = k1 + k2 lambda (mod r) and |k1|, |k2| < 2^129: 128 doublings instead of 252, the additions of
// the two half-length scalars interleaved on one accumulator.  Same group element as groups/mod.rs:250-270, different Jacobian
// coordinates - callers normalize.  Constants and a word-for-word model of glv_decompose: tools/gen_device_constants.py.
struct GlvSplit {
    uint32_t m1[5], m2[5];       // |k1|, |k2| (160-bit little-endian words; < 2^129)
    bool neg1, neg2;
};
// out[0..NO) = low NO words of a[0..NA) * b[0..NB)
template <int NA, int NB, int NO>
BN_FN void words_mul(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    uint32_t t[NA + NB];
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            uint64_t x = (uint64_t)a[i] * b[j] + t[i + j] + c;
            t[i + j] = (uint32_t)x; c = x >> 32;
        }
        t[i + NB] = (uint32_t)c;
    }
#pragma unroll
    for (int i = 0; i < NO; ++i) out[i] = t[i];
}
BN_FN GlvSplit glv_decompose(const uint32_t *k_raw) {
    // c1 = floor(k * floor(2^256 b2 / r) / 2^256),  c2 = floor(k * floor(2^256 |b1| / r) / 2^256)
    uint32_t p1[11], p2[13];
    words_mul<8, 3, 11>(k_raw, k::GLV_G1, p1);
    words_mul<8, 5, 13>(k_raw, k::GLV_G2, p2);
    const uint32_t *c1 = p1 + 8, *c2 = p2 + 8;               // 3 and 5 words
    // modulo 2^192, two's complement:  k1 = k - c1 a1 - c2 a2,   k2 = c1 |b1| - c2 b2
    uint32_t t1[6], t2[6], t3[6], t4[6];
    words_mul<3, 2, 5>(c1, k::GLV_A1, t1); t1[5] = 0;
    words_mul<5, 4, 6>(c2, k::GLV_A2, t2);
    words_mul<3, 4, 6>(c1, k::GLV_B1N, t3);
    words_mul<5, 2, 6>(c2, k::GLV_B2, t4);
    uint32_t k1[6], k2[6];
    int64_t b1 = 0, b2 = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int64_t s = (int64_t)k_raw[i] - (int64_t)t1[i] - (int64_t)t2[i] + b1;
        k1[i] = (uint32_t)s; b1 = s >> 32;
        int64_t u = (int64_t)t3[i] - (int64_t)t4[i] + b2;
        k2[i] = (uint32_t)u; b2 = u >> 32;
    }
    GlvSplit g;
    g.neg1 = (k1[5] >> 31) != 0; g.neg2 = (k2[5] >> 31) != 0;
    uint32_t cy1 = g.neg1 ? 1u : 0u, cy2 = g.neg2 ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 5; ++i) {                             // magnitude: conditional two's-complement negation
        uint64_t a = (uint64_t)(g.neg1 ? ~k1[i] : k1[i]) + cy1; g.m1[i] = (uint32_t)a; cy1 = (uint32_t)(a >> 32);
        uint64_t b = (uint64_t)(g.neg2 ? ~k2[i] : k2[i]) + cy2; g.m2[i] = (uint32_t)b; cy2 = (uint32_t)(b >> 32);
    }
    return g;
}
// radix-16 Booth digit i of a non-negative magnitude (NW words): -8 b[4i+3] + 4 b[4i+2] + 2 b[4i+1] + b[4i] + b[4i-1], in [-8, 8];
// sum digit_i 16^i = magnitude when the window above the top one is empty (no carries to propagate: MSB-first evaluation)
template <int NW = 5>
BN_FN int booth_digit(const uint32_t *mag, int i) {
    const int pos = 4 * i - 1;
    uint32_t x;
    if (pos < 0) {
        x = (mag[0] << 1) & 31u;
    } else {
        const int w = pos >> 5, sh = pos & 31;
        uint64_t two = (uint64_t)mag[w] | (w + 1 < NW ? ((uint64_t)mag[w + 1] << 32) : 0);
        x = (uint32_t)(two >> sh) & 31u;
    }
    return (int)((x >> 1) & 7u) + (int)(x & 1u) - (int)((x >> 4) << 3);
}
constexpr int GLV_WINDOWS = 33;          // 4 * 33 = 132 bits >= 129 + the Booth sign bit
// The same digits as a STREAM, most significant window first: the magnitude is kept shifted so that the five bits of the current window
// are the top bits of its top word - reading a digit is a shift of one register and moving on is NW funnel shifts (v_alignbit).
// booth_digit above indexes the word array with the (wave-uniform but run-time) window number, which the compiler turns into a select
// chain over the registers: 120 instructions per digit in the G1 window loop (round 6: tools/energy_mix.py bn254_g1_mul_ME, run 20).
template <int NW, int WINDOWS>
struct BoothStream {
    uint32_t s[NW];
    BN_FN void init(const uint32_t *mag) {                     // magnitude below 2^(4 WINDOWS - 1): window WINDOWS-1 to the top
        constexpr int SH = 32 * NW - 4 * WINDOWS;
        static_assert(SH > 0 && SH < 32, "BoothStream shift");
#pragma unroll
        for (int i = NW - 1; i >= 1; --i) s[i] = (mag[i] << SH) | (mag[i - 1] >> (32 - SH));
        s[0] = mag[0] << SH;
    }
    static BN_FN int digit_of(uint32_t top_word) {             // -8 b3 + 4 b2 + 2 b1 + b0 + b(-1) of the window in the top five bits
        const uint32_t x = top_word >> 27;
        return (int)((x >> 1) & 7u) + (int)(x & 1u) - (int)((x >> 4) << 3);
    }
    BN_FN int digit() const { return digit_of(s[NW - 1]); }
    BN_FN void next() {
#pragma unroll
        for (int i = NW - 1; i >= 1; --i) s[i] = (s[i] << 4) | (s[i - 1] >> 28);
        s[0] <<= 4;
    }
};

template <class Tab>
BN_FN Jac<FqField> scalar_mul_glv(const Jac<FqField> &p, const uint32_t *k_raw, Tab &aff) {
    using F = FqField;
    const GlvSplit g = glv_decompose(k_raw);
    const bool p_inf = F::is_zero(p.z);
    Jac<F> tab[9];                                            // tab[j] = j P, tab[0] = infinity
    tab[1] = p;
    tab[2] = jac_double(tab[1]);
    tab[4] = jac_double(tab[2]);
    tab[8] = jac_double(tab[4]);
    tab[3] = jac_add_flags(tab[2], p, p_inf, p_inf);
    tab[6] = jac_double(tab[3]);
    tab[5] = jac_add_flags(tab[4], p, p_inf, p_inf);
    tab[7] = jac_add_flags(tab[6], p, p_inf, p_inf);
    const Fe zc = table_to_common_z<F, 8>(tab, aff);          // the table on the isomorphic curve where it is affine
    const Fe beta = fe_const(k::GLV_BETA);
    Jac<F> res = {F::zero(), F::one(), F::zero()};
    bool res_inf = true;
    BoothStream<5, GLV_WINDOWS> d1, d2;
    d1.init(g.m1); d2.init(g.m2);
#pragma unroll 1
    for (int w = GLV_WINDOWS - 1; w >= 0; --w) {
        if (w != GLV_WINDOWS - 1) {
#pragma unroll 1
            for (int d = 0; d < 4; ++d) res = jac_double(res);    // infinity stays infinity (z = 0)
        }
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            BN_G1_HOOK((GLV_WINDOWS - 1 - w) * 2 + half, GLV_WINDOWS * 2);
            const int d = half ? d2.digit() : d1.digit();
            const int ad = d < 0 ? -d : d;
            const bool negate = (d < 0) != (half ? g.neg2 : g.neg1);
            Aff<F> q = aff.get(ad ? ad : 1);                      // digit 0: the operand is ignored (q_inf)
            if (half) q.x = fe_mul(q.x, beta);                    // phi(j P): the endomorphism commutes with the isomorphism
            const bool q_inf = p_inf || ad == 0;
            // the sum of two finite points may be infinity (opposite points), and partial sums of the two interleaved scalars can cancel
            // for crafted inputs - the flag follows the arithmetic (jac_madd_signed), it is not assumed
            res = jac_madd_signed(res, q, negate, res_inf, q_inf);
        }
        d1.next(); d2.next();
    }
    res.z = fe_mul(res.z, zc);                                    // back from the isomorphic curve
    return res;
}

// ---- G2 only: GLS.  psi = twist o Frobenius o untwist - the reference's own mul_by_q (groups/mod.rs:550-555), (x, y) -> (conj(x) gx,
// conj(y) gy) - is multiplication by lambda = q mod r = 6u^2 on the order-r subgroup of the twist, and lambda^4 - lambda^2 + 1 = 0:
// k Q = k0 Q + k1 psi(Q) + k2 psi^2(Q) + k3 psi^3(Q) with |k_i| < 2^67 (Galbraith-Scott; Babai rounding against four short lattice
// vectors in per-lane integer arithmetic, a word-for-word model is asserted over 20 000 scalars in tools/gen_device_constants.py).
// 68 doublings instead of 252; the four Booth digit streams share one accumulator; psi^j is applied to the affine table entry on
// the fly (psi^2 = (w x, -y): one Fq product per coordinate; psi, psi^3: two products by constants).  Same group element as
// groups/mod.rs:250-270, different Jacobian coordinates - callers normalize.
struct GlsSplit {
    uint32_t m[4][3];            // |k_i| (96-bit little-endian words; < 2^67)
    bool neg[4];
};
BN_FN GlsSplit gls_decompose(const uint32_t *k_raw) {
    uint32_t c[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t pr[12];
        words_mul<8, 8, 12>(k_raw, k::GLS_G[j], pr);              // c_j = floor(k G_j / 2^288) mod 2^96
        c[j][0] = pr[9]; c[j][1] = pr[10]; c[j][2] = pr[11];
    }
    GlsSplit g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v[3] = {i == 0 ? k_raw[0] : 0u, i == 0 ? k_raw[1] : 0u, i == 0 ? k_raw[2] : 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t t[3];
            words_mul<3, 3, 3>(c[j], k::GLS_B[j][i], t);
            int64_t cy = 0;
#pragma unroll
            for (int w = 0; w < 3; ++w) {                         // v -= sign_j * t  (mod 2^96)
                int64_t x = (int64_t)v[w] + (k::GLS_GNEG[j] ? (int64_t)t[w] : -(int64_t)t[w]) + cy;
                v[w] = (uint32_t)x; cy = x >> 32;
            }
        }
        g.neg[i] = (v[2] >> 31) != 0;
        uint32_t carry = g.neg[i] ? 1u : 0u;
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            uint64_t x = (uint64_t)(g.neg[i] ? ~v[w] : v[w]) + carry;
            g.m[i][w] = (uint32_t)x; carry = (uint32_t)(x >> 32);
        }
    }
    return g;
}
constexpr int GLS_WINDOWS = 18;          // 4 * 18 = 72 bits >= 67 + the Booth sign bit

template <class F2, class Tab>
BN_FN Jac<Fq2Field<F2>> scalar_mul_gls(const Jac<Fq2Field<F2>> &p, const uint32_t *k_raw, Tab &aff) {
    using F = Fq2Field<F2>;
    const GlsSplit g = gls_decompose(k_raw);
    const bool p_inf = F::is_zero(p.z);
    Jac<F> tab[9];
    tab[1] = p;
    tab[2] = jac_double(tab[1]);
    tab[4] = jac_double(tab[2]);
    tab[8] = jac_double(tab[4]);
    tab[3] = jac_add_flags(tab[2], p, p_inf, p_inf);
    tab[6] = jac_double(tab[3]);
    tab[5] = jac_add_flags(tab[4], p, p_inf, p_inf);
    tab[7] = jac_add_flags(tab[6], p, p_inf, p_inf);
    const F2 zc = table_to_common_z<F, 8, true>(tab, aff);       // entries affine for the common z = norm(zc), an element of Fq
    const F2 zn = f2_mul(zc, f2_conj(zc));
    Jac<F> res = {F::zero(), F::one(), F::zero()};
    bool res_inf = true;
    BoothStream<3, GLS_WINDOWS> ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ds[j].init(g.m[j]);
#pragma unroll 1
    for (int w = GLS_WINDOWS - 1; w >= 0; --w) {
        if (w != GLS_WINDOWS - 1) {
#pragma unroll 1
            for (int d = 0; d < 4; ++d) res = jac_double(res);
        }
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            BN_MUL_HOOK((GLS_WINDOWS - 1 - w) * 4 + j, GLS_WINDOWS * 4);
            const int d = BoothStream<3, GLS_WINDOWS>::digit_of(j == 0 ? ds[0].s[2] : j == 1 ? ds[1].s[2] : j == 2 ? ds[2].s[2] : ds[3].s[2]);
            const int ad = d < 0 ? -d : d;
            const bool negate = ((d < 0) != g.neg[j]) != (j >= 2);            // psi^2, psi^3 carry a minus sign on y
            Aff<F> q = aff.get(ad ? ad : 1);                       // digit 0: the operand is ignored (q_inf)
            if (j == 1) {
                q.x = f2_mul_const(f2_conj_lazy(q.x), k::TWIST_MUL_BY_Q_X); q.y = f2_mul_const(f2_conj_lazy(q.y), k::TWIST_MUL_BY_Q_Y);
            } else if (j == 2) {
                q.x = f2_scale(q.x, f2_scalar_const((const F2 *)nullptr, k::GLS_W));
            } else if (j == 3) {
                q.x = f2_mul_const(f2_conj_lazy(q.x), k::GLS_WGX); q.y = f2_mul_const(f2_conj_lazy(q.y), k::TWIST_MUL_BY_Q_Y);
            }
            q.y = F::select(negate, q.y, F::template lc3<-1, 0, 0>(q.y, q.y, q.y));
            const bool q_inf = p_inf || ad == 0;
            res = jac_madd_flags(res, q, res_inf, q_inf);
            res_inf = F::is_zero_std(res.z);                       // partial sums of the four interleaved parts can cancel
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ds[j].next();
    }
    res.z = F::mul(res.z, zn);                                     // back from the isomorphic curve
    return res;
}

// Gt::pow with the Frobenius endomorphism (Galbraith-Scott in the target group).  pi(f) = f^q, and on elements of ORDER r the
// exponent only counts mod r, where q = lambda = 6u^2 - the same eigenvalue psi has on G2 (e(P, psi Q) = e(P, Q)^q).  So with the
// decomposition of scalar_mul_gls, f^k = prod_i pi^i(f)^(k_i), |k_i| < 2^67: 68 cyclotomic squarings instead of 252, the four Booth
// digit streams share the accumulator (72 products), conj() is the inverse.  The table holds one, then a^1 .. a^8 under pi^0 .. pi^3
// (a Frobenius map is five Fq2 products by constants: 24 of them here instead of 54 on the fly).
// PRECONDITION: base has order dividing r - true of every value the reference's API can produce (lib.rs:165-183: Gt::one, pairing(),
// products, powers and inverses of such; the Fq12 inside Gt is private and Gt has no decoder).  The caller has checked that base is
// cyclotomic, which every such value is; order r itself would cost an exponentiation to check - the strict mode of the kernel
// (gt_pow_cyclotomic) is exact for ANY cyclotomic element instead.
constexpr int GT_GLS_ENTRIES = 1 + 4 * 8;
// the table: one, a^1 .. a^8 and their three Frobenius images (33 entries).  A function of its own so that a kernel can keep it OUT of
// line: inlined, its values stayed live into the window loop's register allocation (17-22 spilled VGPRs in bn254_gt_pow_B)
template <class F2, class Tbl>
BN_FN void gt_pow_gls_table(const Fq12<F2> &base, Tbl &tbl) {
    tbl.put(0, f12_one<F2>());
    tbl.put(1, base);
#pragma unroll 1
    for (int e = 2; e <= 8; ++e) {                 // a^e = (a^(e/2))^2 for even e, a^(e-1) * a for odd e
        Fq12<F2> v;
        if ((e & 1) == 0) v = f12_cyclotomic_sqr(Fq12<F2>{tbl.c0(e >> 1), tbl.c1(e >> 1)});
        else v = f12_mul_src(Fq12<F2>{tbl.c0(e - 1), tbl.c1(e - 1)}, Fq12Slot<F2, Tbl>{tbl, 1}, false);
        tbl.put(e, v);
    }
#pragma unroll 1
    for (int e = 1; e <= 8; ++e) {
        const Fq12<F2> v = {tbl.c0(e), tbl.c1(e)};
        tbl.put(8 + e, f12_frobenius_one<1>(v));
        tbl.put(16 + e, f12_frobenius_one<2>(v));
        tbl.put(24 + e, f12_frobenius_one<3>(v));
    }
}
// the window loop over a table built by gt_pow_gls_table
template <class F2, class Tbl>
BN_FN Fq12<F2> gt_pow_gls_loop(const uint32_t *k_raw, Tbl &tbl) {
    const GlsSplit g = gls_decompose(k_raw);
    Fq12<F2> res = f12_one<F2>();
#pragma unroll 1
    for (int w = GLS_WINDOWS - 1; w >= 0; --w) {
        if (w != GLS_WINDOWS - 1) {
#pragma unroll 1
            for (int d = 0; d < 4; ++d) res = f12_cyclotomic_sqr(res);
        }
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            BN_EXP_HOOK((GLS_WINDOWS - 1 - w) * 4 + j, GLS_WINDOWS * 4);           // the two resident waves' hand-over (bn254_kernels_b.hip; a no-op elsewhere)
            // (the digit STREAM of scalar_mul_gls is neutral here - 18.0 M pows/s either way - and costs the register allocation of the product
            //  blocks 11 more spilled VGPRs: profiles/r06_ab_gtpow_mapped_table.txt)
            const int d = booth_digit<3>(g.m[j], w);
            const int ad = d < 0 ? -d : d;
            const bool neg = (d < 0) != g.neg[j];
            // res * conj(t) = conj(conj(res) * t): the sign goes on the running value, the table operand takes the plain path
            res.c1 = f6_cond_neg(neg, res.c1);
            // (forming the Frobenius image when an entry is USED instead of storing 24 images per element - 9 table entries instead of 33, 0.75 GB of
            //  stores less per 2^16 - was built and measured 11 % SLOWER: 54 maps of five Fq2 products by constants per element against 24, and the
            //  stores were never the cost.  profiles/r06_ab_gtpow_mapped_table.txt)
            res = f12_mul_src(res, Fq12Slot<F2, Tbl>{tbl, ad ? 8 * j + ad : 0}, false);
            res.c1 = f6_cond_neg(neg, res.c1);
        }
    }
    return res;
}
template <class F2, class Tbl>
BN_FN Fq12<F2> gt_pow_gls(const Fq12<F2> &base, const uint32_t *k_raw, Tbl &tbl) {
    gt_pow_gls_table(base, tbl);
    return gt_pow_gls_loop<F2>(k_raw, tbl);
}

// the same chains with the table in a local array (host simulation)
BN_FN Jac<FqField> scalar_mul_glv(const Jac<FqField> &p, const uint32_t *k_raw) { AffTableVars<FqField> t; return scalar_mul_glv(p, k_raw, t); }
template <class F2> BN_FN Jac<Fq2Field<F2>> scalar_mul_gls(const Jac<Fq2Field<F2>> &p, const uint32_t *k_raw) { AffTableVars<Fq2Field<F2>> t; return scalar_mul_gls<F2>(p, k_raw, t); }
template <class F> BN_FN Jac<F> scalar_mul_booth_affine(const Jac<F> &p, const uint32_t *k_raw) { AffTableVars<F> t; return scalar_mul_booth_affine<F>(p, k_raw, t); }

// lib.rs:88-95 (normalize): (x/z^2, y/z^3, 1).  Infinity is returned as G::zero() = (0, 1, 0) (groups/mod.rs:208-214): that is
// what the reference holds for every valid input that multiplies to zero (k = 0 or p = 0; k < r excludes the rest), while
// the windowed chain above may reach z = 0 with other x, y.
template <class F>
BN_FN Jac<F> jac_normalize(const Jac<F> &p) {
    using T = typename F::T;
    bool inf = F::is_zero(p.z);
    T zi = F::inverse(p.z), zi2 = F::sqr(zi);
    Jac<F> r = {F::mul(p.x, zi2), F::mul(p.y, F::mul(zi2, zi)), F::one()};
    r.x = F::select(inf, r.x, F::zero()); r.y = F::select(inf, r.y, F::one()); r.z = F::select(inf, r.z, F::zero());
    return r;
}

}  // namespace bn254

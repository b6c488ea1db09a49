// Fq6 = Fq2[v]/(v^3 - xi) and Fq12 = Fq6[w]/(w^2 - v) for the HIP pairing engine, generic over the Fq2 lane mapping
// (Fq2A / Fq2B, fq2.hpp).  Formulas follow the reference op for op where the reference fixes the VALUE
// (src/fields/fq6.rs, src/fields/fq12.rs); any rearrangement below is value-preserving and the results are compared
// bit for bit with the oracle (tests/test_hostsim.py on CPU, tests/test_gpu_parity.py on the GPU).
//
// Every function takes and returns components in standard form S (normalized limbs, value < 3q) - see fq2.hpp.  Lazy
// intermediates (plain limb-wise sums/differences) are folded back by the fused f2_lc3 / f2_lc_xi reductions.
#pragma once
#include "fq2.hpp"

namespace bn254 {

template <class F2> struct Fq6 { F2 c0, c1, c2; };
template <class F2> struct Fq12 { Fq6<F2> c0, c1; };

#define F2P ((const F2 *)nullptr)

// ------------------------------------------------------------------------------------------------------------- Fq6
template <class F2> BN_FN Fq6<F2> f6_zero() { return {f2_zero(F2P), f2_zero(F2P), f2_zero(F2P)}; }
template <class F2> BN_FN Fq6<F2> f6_one() { return {f2_one(F2P), f2_zero(F2P), f2_zero(F2P)}; }
// reduce(C1*x + C2*y + C3*z) componentwise
template <int C1, int C2, int C3, class F2>
BN_FN Fq6<F2> f6_lc3(const Fq6<F2> &x, const Fq6<F2> &y, const Fq6<F2> &z) {
    return {f2_lc3<C1, C2, C3>(x.c0, y.c0, z.c0), f2_lc3<C1, C2, C3>(x.c1, y.c1, z.c1), f2_lc3<C1, C2, C3>(x.c2, y.c2, z.c2)};
}
// a + b in the cheapest form the FIRST operand of f6_mul accepts (lane-pair mapping: carries propagated, not reduced)
template <class F2> BN_FN Fq6<F2> f6_add_norm(const Fq6<F2> &a, const Fq6<F2> &b) {
    return {f2_sum_for_mul(a.c0, b.c0), f2_sum_for_mul(a.c1, b.c1), f2_sum_for_mul(a.c2, b.c2)};
}
template <class F2> BN_FN Fq6<F2> f6_neg(const Fq6<F2> &a) { return f6_lc3<-1, 0, 0>(a, a, a); }

// fq6.rs:144-158: 6 Fq2 products (Karatsuba), the two xi-multiplications folded into the final reductions; each output
// coefficient is finished as soon as its products exist (short live ranges: see f12_mul_by_024)
template <class F2>
BN_COARSE Fq6<F2> f6_mul(const Fq6<F2> &a, const Fq6<F2> &b) {
    F2 aa = f2_mul(a.c0, b.c0), bb = f2_mul(a.c1, b.c1), cc = f2_mul(a.c2, b.c2);
    Fq6<F2> r;
    r.c0 = f2_lc_xi<1, 1>(f2_cross(a.c1, a.c2, b.c1, b.c2, bb, cc), aa);          // xi (a1 b2 + a2 b1) + a0 b0
    r.c1 = f2_lc_xi<1, 1>(cc, f2_cross(a.c0, a.c1, b.c0, b.c1, aa, bb));          // xi a2 b2 + a0 b1 + a1 b0
    r.c2 = f2_lc3<1, 1, 0>(f2_cross(a.c0, a.c2, b.c0, b.c2, aa, cc), bb, bb);     // a0 b2 + a2 b0 + a1 b1
    return r;
}
// The six Karatsuba products of f6_mul WITHOUT their recombination: a caller that subtracts or adds further Fq6 values right away
// (the cross term of an Fq12 product or square) folds everything into ONE fused reduction per coefficient (f2_lc_xi2w / f2_lc3sw) instead
// of three reductions here and three more there (round 4: Miller 3.806 -> 3.765 ms, final exponentiation 3.357 -> 3.341 ms, Gt::pow + 2.3 %;
// profiles/r04g_ab_merged_recombination.txt).
//   a b = (xi x12 + v0) + (xi v2 + x01) v + (x02 + v1) v^2,   x_ij = a_i b_j + a_j b_i (f2_cross: signed lazy sums, limb bound 3)
template <class F2> struct Fq6Raw { F2 v0, v1, v2, x12, x01, x02; };
template <class F2>
BN_COARSE Fq6Raw<F2> f6_mul_raw(const Fq6<F2> &a, const Fq6<F2> &b) {
    Fq6Raw<F2> r;
    r.v0 = f2_mul(a.c0, b.c0); r.v1 = f2_mul(a.c1, b.c1); r.v2 = f2_mul(a.c2, b.c2);
    r.x12 = f2_cross(a.c1, a.c2, b.c1, b.c2, r.v1, r.v2);
    r.x01 = f2_cross(a.c0, a.c1, b.c0, b.c1, r.v0, r.v1);
    r.x02 = f2_cross(a.c0, a.c2, b.c0, b.c2, r.v0, r.v2);
    return r;
}
// fq6.rs:113-127 (CH-SQR2)
template <class F2>
BN_COARSE Fq6<F2> f6_sqr(const Fq6<F2> &a) {
    F2 s0 = f2_sqr(a.c0), ab = f2_mul(a.c0, a.c1);
    F2 s2 = f2_sqr(f2_lc3<1, -1, 1>(a.c0, a.c1, a.c2));
    F2 bc = f2_mul(a.c1, a.c2), s4 = f2_sqr(a.c2);
    F2 s1 = f2_dbl(ab), s3 = f2_dbl(bc);
    Fq6<F2> r;
    r.c0 = f2_lc_xi<1, 1>(s3, s0);
    r.c1 = f2_lc_xi<1, 1>(s4, s1);
    r.c2 = f2_lc3w<1, -1, -1>(f2_add(f2_add(s1, s2), s3), s0, s4);
    return r;
}
template <class F2> BN_FN Fq6<F2> f6_scale(const Fq6<F2> &a, const F2 &by) { return {f2_mul(a.c0, by), f2_mul(a.c1, by), f2_mul(a.c2, by)}; }
// fq6.rs:75-81, P in {1,2,3}
template <int P, class F2>
BN_FN Fq6<F2> f6_frobenius(const Fq6<F2> &a) {
    if constexpr (P % 2 == 0) {
        return {a.c0, f2_mul_const(a.c1, k::FROB6_C1[P]), f2_mul_const(a.c2, k::FROB6_C2[P])};
    } else {
        return {f2_conj(a.c0), f2_mul_const(f2_conj_lazy(a.c1), k::FROB6_C1[P]), f2_mul_const(f2_conj_lazy(a.c2), k::FROB6_C2[P])};
    }
}
// fq6.rs:129-141
template <class F2>
BN_OUTER Fq6<F2> f6_inverse(const Fq6<F2> &a) {
    F2 c0 = f2_lc3<1, -1, 0>(f2_sqr(a.c0), f2_mul(a.c1, f2_mul_xi(a.c2)), a.c0);
    F2 c1 = f2_lc_xi<1, -1>(f2_sqr(a.c2), f2_mul(a.c0, a.c1));
    F2 c2 = f2_lc3<1, -1, 0>(f2_sqr(a.c1), f2_mul(a.c0, a.c2), a.c0);
    F2 n = f2_lc_xi<1, 1>(f2_add(f2_mul(a.c2, c1), f2_mul(a.c1, c2)), f2_mul(a.c0, c0));
    F2 t = f2_inverse(n);
    return {f2_mul(t, c0), f2_mul(t, c1), f2_mul(t, c2)};
}

// ------------------------------------------------------------------------------------------------------------- Fq12
template <class F2> BN_FN Fq12<F2> f12_one() { return {f6_one<F2>(), f6_zero<F2>()}; }
// fq12.rs:295-307.  `b` is read through a source object (b.c0(), b.c1() hand out its halves on demand) so that a multiplier
// that stays constant across a loop can live in LDS instead of 54 VGPRs; `conj_b` multiplies by the conjugate (c0, -c1).
template <class F2>
struct Fq12Ref {
    const Fq12<F2> &x;
    BN_FN Fq6<F2> c0() const { return x.c0; }
    BN_FN Fq6<F2> c1() const { return x.c1; }
};
template <class F2, class BSrc>
BN_FN Fq12<F2> f12_mul_src(const Fq12<F2> &a, const BSrc &b, bool conj_b) {
    // the Karatsuba cross term first: it is the only product that needs both halves of a and of b at once
    Fq6<F2> b1 = b.c1();
    if (conj_b) b1 = f6_neg(b1);
    const Fq6Raw<F2> t = f6_mul_raw(f6_add_norm(a.c0, a.c1), f6_add_norm(b.c0(), b1));
    Fq6<F2> bb = f6_mul(a.c1, b1);
    Fq6<F2> aa = f6_mul(a.c0, b.c0());
    Fq12<F2> r;
    // s t - aa - bb with the recombination of s t folded in
    r.c1.c0 = f2_lc_xi2w<1, 1, -1>(t.x12, f2_ssub(t.v0, aa.c0), bb.c0);
    r.c1.c1 = f2_lc_xi2w<1, 1, -1>(t.v2, t.x01, f2_add(aa.c1, bb.c1));
    r.c1.c2 = f2_lc3sw<1, -1, 0>(f2_add(t.x02, t.v1), f2_add(aa.c2, bb.c2), aa.c2);
    r.c0.c0 = f2_lc_xi<1, 1>(bb.c2, aa.c0);                  // aa + v*bb
    r.c0.c1 = f2_lc3<1, 1, 0>(aa.c1, bb.c0, bb.c0);
    r.c0.c2 = f2_lc3<1, 1, 0>(aa.c2, bb.c1, bb.c1);
    return r;
}
template <class F2>
BN_COARSE Fq12<F2> f12_mul(const Fq12<F2> &a, const Fq12<F2> &b) { return f12_mul_src(a, Fq12Ref<F2>{b}, false); }
// fq12.rs:275-282 (complex squaring over Fq6)
// REDUCED_C1: return c1 = 2ab as a fused reduction (value < 2q) instead of a carry-propagated sum (< 4q).  The Miller loop feeds
// every square straight into a sparse product, which takes the cheaper form; chains of squarings (Gt::pow) need the reduced one -
// with c1 < 4q the Karatsuba sums of the NEXT squaring can exceed the 9q bias of the lane-pair product (found by the host
// simulation's bound checks; actual values stay far below, but the bound must hold by construction).
template <bool REDUCED_C1 = false, class F2>
BN_COARSE Fq12<F2> f12_sqr(const Fq12<F2> &a) {
    Fq6<F2> ab = f6_mul(a.c0, a.c1);
    Fq6<F2> u;                                                // v*c1 + c0
    u.c0 = f2_lc_xi<1, 1>(a.c1.c2, a.c0.c0);
    u.c1 = f2_sum_for_mul(a.c1.c0, a.c0.c1);
    u.c2 = f2_sum_for_mul(a.c1.c1, a.c0.c2);
    Fq12<F2> r;
    // t - ab - v ab with the recombination of t = (c0 + c1) u folded in: three reductions instead of six
    const Fq6Raw<F2> t = f6_mul_raw(f6_add_norm(a.c0, a.c1), u);
    r.c0.c0 = f2_lc_xi2w<1, 1, -1>(f2_ssub(t.x12, ab.c2), t.v0, ab.c0);
    r.c0.c1 = f2_lc_xi2w<1, 1, -1>(t.v2, t.x01, f2_add(ab.c1, ab.c0));
    r.c0.c2 = f2_lc3sw<1, -1, 0>(f2_add(t.x02, t.v1), f2_add(ab.c2, ab.c1), ab.c1);
    if constexpr (REDUCED_C1) r.c1 = f6_lc3<2, 0, 0>(ab, ab, ab);
    else r.c1 = f6_add_norm(ab, ab);                           // 2ab as a plain sum (carries propagated)
    return r;
}
// fq12.rs:103-105
template <class F2> BN_OUTER Fq12<F2> f12_conj(const Fq12<F2> &a) { return {a.c0, f6_neg(a.c1)}; }
// fq12.rs:284-292
template <class F2>
BN_OUTER Fq12<F2> f12_inverse(const Fq12<F2> &a) {
    Fq6<F2> s1 = f6_sqr(a.c1);
    Fq6<F2> s0 = f6_sqr(a.c0);
    Fq6<F2> d;                                                // c0^2 - v*c1^2
    d.c0 = f2_lc_xi<-1, 1>(s1.c2, s0.c0);
    d.c1 = f2_lc3<1, -1, 0>(s0.c1, s1.c0, s1.c0);
    d.c2 = f2_lc3<1, -1, 0>(s0.c2, s1.c1, s1.c1);
    Fq6<F2> t = f6_inverse(d);
    return {f6_mul(a.c0, t), f6_neg(f6_mul(a.c1, t))};
}
// fq12.rs:90-95, P in {1,2,3}
template <int P, class F2>
BN_OUTER Fq12<F2> f12_frobenius(const Fq12<F2> &a) {
    Fq6<F2> c1 = f6_frobenius<P>(a.c1);
    F2 g = f2_const(F2P, k::FROB12_C1[P]);
    return {f6_frobenius<P>(a.c0), f6_scale(c1, g)};
}

// The same map with ONE constant per coefficient: the reference maps c1 through the Fq6 map and then scales it by FROB12_C1 - seven
// Fq2 products; with the products of the constants tabulated it is five, and for P = 2 all five lie in Fq - a scaling of each
// component.  Same field elements, hence the same bytes.  Used by Gt::pow (24 maps per element).  NOT by the final-exponentiation
// kernel: replacing its four maps moved the register allocation of that kernel's hot loops (15-22 spills per step appeared in the
// squaring and product blocks, 3.40 -> 3.96 ms: profiles/r03y_ab_frobenius.txt) - the compiler allocates across these calls.
template <int P, class F2>
BN_OUTER Fq12<F2> f12_frobenius_one(const Fq12<F2> &a) {
    if constexpr (P == 2) {
        auto sc = [&](const F2 &x, int i) { return f2_scale(x, f2_scalar_const(F2P, k::FROB2_S[i])); };
        return {{a.c0.c0, sc(a.c0.c1, 0), sc(a.c0.c2, 1)}, {sc(a.c1.c0, 2), sc(a.c1.c1, 3), sc(a.c1.c2, 4)}};
    } else {
        return {f6_frobenius<P>(a.c0),
                {f2_mul_const(f2_conj_lazy(a.c1.c0), k::FROB12_C1[P]), f2_mul_const(f2_conj_lazy(a.c1.c1), k::FROB12_C1C1[P]),
                 f2_mul_const(f2_conj_lazy(a.c1.c2), k::FROB12_C1C2[P])}};
    }
}

// fq12.rs:107-176: f * (x0 + x2 v^2 + x4 v w), 13 Fq2 products; (ell_0, ell_vw, ell_vv) -> (x0, x4, x2) as in the reference.
// Same 13 products and the same sums as the reference, but ordered so that every product dies as early as possible (each
// output coefficient is finished as soon as its inputs exist): the live set stays inside the 256-VGPR budget of a wave that
// shares its SIMD, instead of parking a dozen 9-register products in private memory.
template <class F2>
BN_COARSE Fq12<F2> f12_mul_by_024(const Fq12<F2> &f, const F2 &ell_0, const F2 &ell_vw, const F2 &ell_vv) {
    const F2 &z0 = f.c0.c0, &z1 = f.c0.c1, &z2 = f.c0.c2, &z3 = f.c1.c0, &z4 = f.c1.c1, &z5 = f.c1.c2;
    const F2 &x0 = ell_0, &x2 = ell_vv, &x4 = ell_vw;
    Fq12<F2> r;
    // Register-pressure order (the spills of this function cost the Miller loop ~10 %, profiles/r02k_*): the four Karatsuba sums
    // of f's coefficients are formed FIRST, so that every z_i dies right after its own products instead of living to the end;
    // peak ~17 nine-register values instead of ~21.
    const F2 z135 = f2_sum3_for_mul(z1, z3, z5);
    const F2 z02 = f2_add(z0, z2), z24 = f2_add(z2, z4), z04 = f2_add(z0, z4);
    BN_COMPILER_FENCE();
    const F2 d0 = f2_mul(z0, x0), d2 = f2_mul(z2, x2), d4 = f2_mul(z4, x4);          // z0, z2, z4 dead
    F2 s1;                                                                           // running sum of the six cross products (lazy)
    F2 z1x0;
    {
        F2 z1x2 = f2_mul(z1, x2);
        z1x0 = f2_mul(z1, x0);                                                       // z1 dead
        r.c0.c0 = f2_lc_xi<1, 1>(f2_add(z1x2, d4), d0);                              // xi (z1 x2 + z4 x4) + z0 x0
        s1 = f2_add(z1x2, z1x0);
    }
    BN_COMPILER_FENCE();
    F2 z5x2;
    {
        F2 z5x4 = f2_mul(z5, x4);
        z5x2 = f2_mul(z5, x2);                                                       // z5 dead
        r.c0.c1 = f2_lc_xi<1, 1>(f2_add(z5x4, d2), z1x0);                            // xi (z5 x4 + z2 x2) + z1 x0
        s1 = f2_add(f2_add(s1, z5x4), z5x2);
    }
    BN_COMPILER_FENCE();
    F2 z3x0;
    {
        F2 z3x4 = f2_mul(z3, x4);
        z3x0 = f2_mul(z3, x0);                                                       // z3 dead
        s1 = f2_add(f2_add(s1, z3x4), z3x0);
        F2 m02 = f2_mul(z02, f2_norm(f2_add(x0, x2)));
        r.c0.c2 = f2_lc3<1, -1, -1>(f2_add(m02, z3x4), d0, d2);                      // (z0+z2)(x0+x2) - d0 - d2 + z3 x4
    }
    BN_COMPILER_FENCE();
    {
        F2 m24 = f2_mul(z24, f2_norm(f2_add(x2, x4)));
        r.c1.c0 = f2_lc_xi<1, 1>(f2_ssub(f2_ssub(m24, d2), d4), z3x0);               // xi ((z2+z4)(x2+x4) - d2 - d4) + z3 x0
    }
    {
        F2 m04 = f2_mul(z04, f2_norm(f2_add(x0, x4)));
        r.c1.c1 = f2_lc_xi<1, 1>(z5x2, f2_ssub(f2_ssub(m04, d0), d4));               // xi z5 x2 + (z0+z4)(x0+x4) - d0 - d4
    }
    BN_COMPILER_FENCE();
    F2 ms = f2_mul(z135, f2_sum3_for_mul(x0, x2, x4));
    r.c1.c2 = f2_lc3w<1, -1, 0>(ms, s1, s1);                                          // (z1+z3+z5)(x0+x2+x4) - s1
    return r;
}

// The same product in the lane-pair mapping by LAZY REDUCTION instead of Karatsuba: each of the six output coefficients is a sum of three
// Fq2 products,
//     c0' = (z0 x0 + xi z1 x2 + xi z4 x4) + (z1 x0 + xi z2 x2 + xi z5 x4) v + (z2 x0 + z0 x2 + z3 x4) v^2
//     c1' = (z3 x0 + xi z4 x2 + xi z2 x4) + (z4 x0 + xi z5 x2 + z0 x4) v + (z5 x0 + z3 x2 + z1 x4) v^2
// and the three share ONE Montgomery reduction per lane (fe_mul6: 486 + 81 multiply-adds).  18 products instead of 13, but six reductions
// instead of thirteen, no recombination at all (the Karatsuba form needs six fused reductions, ~95 instructions each), and every
// multiplier is prepared once (x0, x2, x4, xi x2, xi x4): 3402 multiply-adds and ~4.2 k instructions per lane against 3159 and ~5.1 k.
// On this machine an instruction that is not a multiply-add costs about as much issue time as one that is, so fewer instructions win.
template <class T>
BN_COARSE Fq12<Fq2B<T>> f12_mul_by_024(const Fq12<Fq2B<T>> &f, const Fq2B<T> &ell_0, const Fq2B<T> &ell_vw, const Fq2B<T> &ell_vv) {
    typedef Fq2B<T> F2;
    const F2 &z0 = f.c0.c0, &z1 = f.c0.c1, &z2 = f.c0.c2, &z3 = f.c1.c0, &z4 = f.c1.c1, &z5 = f.c1.c2;
    Fq12<F2> r;
    const Fq2BPrep<T> x0 = f2b_prepare(ell_0);
    {
        const Fq2BPrep<T> xx2 = f2b_prepare(f2_mul_xi(ell_vv));
        {
            const Fq2BPrep<T> xx4 = f2b_prepare(f2_mul_xi(ell_vw));
            r.c0.c0 = f2b_mul3(z0, x0, z1, xx2, z4, xx4);
            r.c0.c1 = f2b_mul3(z1, x0, z2, xx2, z5, xx4);
            r.c1.c0 = f2b_mul3(z3, x0, z4, xx2, z2, xx4);
        }
        BN_COMPILER_FENCE();
        const Fq2BPrep<T> x4 = f2b_prepare(ell_vw);
        r.c1.c1 = f2b_mul3(z4, x0, z5, xx2, z0, x4);
        BN_COMPILER_FENCE();
        const Fq2BPrep<T> x2 = f2b_prepare(ell_vv);
        r.c0.c2 = f2b_mul3(z2, x0, z0, x2, z3, x4);
        r.c1.c2 = f2b_mul3(z5, x0, z3, x2, z1, x4);
    }
    return r;
}

// The sparse line product of the NATIVE prepared-G2 mode (pairing.hpp miller_loop_native).  A line ell_0 + (ell_vw y_P) v w + (ell_vv x_P) v^2
// may be scaled by any element of a proper subfield of Fq12 - the final exponentiation kills it - so the prepared table holds every line
// divided by ell_vw (Q only, one inversion per table) and the kernel divides by x_P (P only, the pairing's one inversion):
//     line' = (A sigma) + tau v w + B v^2,     A = ell_0 / ell_vw,  B = ell_vv / ell_vw  in Fq2 (table),   sigma = 1 / x_P,  tau = y_P / x_P  in Fq
// With the lazily reduced form of f12_mul_by_024 above (x0 = A sigma, x2 = B, x4 = tau):
//     c0' = (z0 x0 + xi z1 B + xi z4 tau) + (z1 x0 + xi z2 B + xi z5 tau) v + (z2 x0 + z0 B + z3 tau) v^2
//     c1' = (z3 x0 + xi z4 B + xi z2 tau) + (z4 x0 + xi z5 B + z0 tau) v + (z5 x0 + z3 B + z1 tau) v^2
// B and xi B arrive PREPARED from the table (no conversion, no xi-multiplication, no operand set-up per pairing), a product by tau is 81
// instead of 162 multiply-adds per lane (fe_mul5), xi z tau is the dual product own * (9 tau) + partner * (-+ tau) with both multipliers
// made once per pairing, and the ONE per-line scaling left is x0 = A sigma (one product per lane + the operand set-up).  2673 + 486 + 171
// multiply-adds per line and lane against 3402 + 342 + the conversions of the reference-image coefficients (miller_loop_prepared).
//   src.x0(): prepared A sigma;  src.xb() / src.b(): prepared xi B / B;  src.tau(), src.tau9() (9 tau, normalized), src.taum() (-tau on the even
//   lane, tau on the odd lane)
template <class T, class Src>
BN_COARSE Fq12<Fq2B<T>> f12_mul_by_line_native(const Fq12<Fq2B<T>> &f, const Src &src) {
    typedef Fq2B<T> F2;
    const F2 &z0 = f.c0.c0, &z1 = f.c0.c1, &z2 = f.c0.c2, &z3 = f.c1.c0, &z4 = f.c1.c1, &z5 = f.c1.c2;
    Fq12<F2> r;
    const Fq2BPrep<T> x0 = src.x0();
    {
        const Fq2BPrep<T> xb = src.xb();
        {
            const Fq2BPrep<T> xt = {src.tau9(), src.taum()};
            r.c0.c0 = f2b_mul3(z0, x0, z1, xb, z4, xt);
            r.c0.c1 = f2b_mul3(z1, x0, z2, xb, z5, xt);
            r.c1.c0 = f2b_mul3(z3, x0, z4, xb, z2, xt);
        }
        BN_COMPILER_FENCE();
        r.c1.c1 = f2b_mul3s(z4, x0, z5, xb, z0, src.tau());
    }
    BN_COMPILER_FENCE();
    const Fq2BPrep<T> b = src.b();
    const T tau = src.tau();
    r.c0.c2 = f2b_mul3s(z2, x0, z0, b, z3, tau);
    r.c1.c2 = f2b_mul3s(z5, x0, z3, b, z1, tau);
    return r;
}

// one Fp4 squaring of Granger-Scott, fused with the "times three, plus/minus twice the old coefficient" that follows it:
//   tmp = a b,  t_even = (a + b)(a + xi b) - tmp - xi tmp = a^2 + xi b^2
//   out_even = 3 t_even - 2 z_even          (ONE reduction: 3 m - 3 tmp - 3 xi tmp - 2 z_even)
//   returns tmp (t_odd = 2 tmp is applied by the caller: 6 tmp + 2 z_odd)
template <class F2>
BN_FN F2 f4_sq_fused(const F2 &a, const F2 &b, const F2 &z_even, F2 &out_even) {
    F2 tmp = f2_mul(a, b);
    F2 m = f2_mul(f2_add(a, b), f2_lc_xi<1, 1>(b, a));
    out_even = f2_lc_xi2<-3, 3, -2>(tmp, f2_ssub(m, tmp), z_even);
    return tmp;
}
template <class F2>
BN_COARSE Fq12<F2> f12_cyclotomic_sqr(const Fq12<F2> &f) {
    const F2 &z0 = f.c0.c0, &z4 = f.c0.c1, &z3 = f.c0.c2, &z2 = f.c1.c0, &z1 = f.c1.c1, &z5 = f.c1.c2;
    Fq12<F2> r;
    {
        F2 p01 = f4_sq_fused(z0, z1, z0, r.c0.c0);           // z0' = 2(t0 - z0) + t0
        r.c1.c1 = f2_lc3<6, 2, 0>(p01, z1, z1);               // z1' = 2(t1 + z1) + t1,  t1 = 2 p01
    }
    {
        F2 p23 = f4_sq_fused(z2, z3, z4, r.c0.c1);           // z4' = 2(t2 - z4) + t2
        r.c1.c2 = f2_lc3<6, 2, 0>(p23, z5, z5);               // z5' = 2(t3 + z5) + t3
    }
    {
        F2 p45 = f4_sq_fused(z4, z5, z3, r.c0.c2);           // z3' = 2(t4 - z3) + t4
        r.c1.c0 = f2_lc_xi<6, 2>(p45, z2);                    // z2' = 2(xi t5 + z2) + xi t5
    }
    return r;
}

// out-of-line twins for the straight-line (non-loop) callers
template <class F2> BN_OUTER Fq12<F2> f12_mul_o(const Fq12<F2> &a, const Fq12<F2> &b) { return f12_mul(a, b); }

#undef F2P
}  // namespace bn254

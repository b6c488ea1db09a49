// Fq2 = Fq[i]/(i^2+1) for the HIP pairing engine, two interchangeable lane mappings behind one set of f2_* functions
// (the tower and the pairing in tower.hpp / pairing.hpp are templates over the Fq2 type):
//
//   Fq2A  one lane holds both components (c0,c1): Karatsuba, 3 Montgomery products + recombination.
//   Fq2B  a PAIR of adjacent lanes holds one element: even lane c0, odd lane c1.  A product is
//             c0 = a0*b0 + a1*(-b1)       c1 = a0*b1 + a1*b0
//         i.e. every lane evaluates own_a*u + partner_a*v with ONE Montgomery reduction (fe_mul2, 243 mads): the same
//         multiplier work as Karatsuba, half the per-lane state (the whole Miller loop then lives in VGPRs + LDS), half
//         the additions, and the result is already in standard form.  Partner limbs arrive by DPP quad_perm [1,0,3,2].
//
// Reference semantics: src/fields/fq2.rs (mul :139-155, squared :112-123, scale :63-68, mul_by_nonresidue :70-72,
// frobenius_map :74-83, inverse :125-136).  The reference spends a 4th Montgomery product on beta = -1 and a full Fq2
// product on xi = 9+i; here both are additions.
//
// "Standard form" S = normalized limbs, value < 3q.  Contract of every f2_* function unless stated otherwise:
// inputs S, outputs S.  f2_mul additionally accepts a lazy first operand (lb <= 2, vb <= 6).
#pragma once
#include "fe.hpp"

namespace bn254 {

// ===================================================================================================== policy A
struct Fq2A {
    Fe c0, c1;
    using Scalar = Fe;
};

BN_FN Fq2A f2_add(const Fq2A &a, const Fq2A &b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
BN_FN Fq2A f2_dbl(const Fq2A &a) { return f2_add(a, a); }
template <int LB, int K>
BN_FN Fq2A f2_sub(const Fq2A &a, const Fq2A &b) { return {fe_sub<LB, K>(a.c0, b.c0), fe_sub<LB, K>(a.c1, b.c1)}; }
BN_FN Fq2A f2_norm(const Fq2A &a) { return {fe_norm(a.c0), fe_norm(a.c1)}; }
BN_FN Fq2A f2_std(const Fq2A &a) { return {fe_std(a.c0), fe_std(a.c1)}; }
// componentwise reduce(C1*x + C2*y + C3*z)
template <int C1, int C2, int C3>
BN_FN Fq2A f2_lc3(const Fq2A &x, const Fq2A &y, const Fq2A &z) {
    return {fe_lc3<C1, C2, C3>(x.c0, y.c0, z.c0), fe_lc3<C1, C2, C3>(x.c1, y.c1, z.c1)};
}
BN_FN Fq2A f2_neg(const Fq2A &a) { return f2_lc3<-1, 0, 0>(a, a, a); }
BN_FN Fq2A f2_conj(const Fq2A &a) { return {a.c0, fe_lc3<-1, 0, 0>(a.c1, a.c1, a.c1)}; }
BN_FN Fq2A f2_zero(const Fq2A *) { return {fe_zero(), fe_zero()}; }
BN_FN Fq2A f2_one(const Fq2A *) { return {fe_one(), fe_zero()}; }
template <class T>
BN_FN Fq2A f2_const(const Fq2A *, const T &tab) { return {fe_const(tab[0]), fe_const(tab[1])}; }
BN_FN bool f2_is_zero(const Fq2A &a) { return fe_is_zero(a.c0) & fe_is_zero(a.c1); }
BN_FN Fq2A f2_select(bool take_b, const Fq2A &a, const Fq2A &b) { return {fe_select(take_b, a.c0, b.c0), fe_select(take_b, a.c1, b.c1)}; }

// a: lb <= 2, vb <= 6;  b: S
BN_FN Fq2A f2_mul(const Fq2A &a, const Fq2A &b) {
    Fe aa = fe_mul(a.c0, b.c0), bb = fe_mul(a.c1, b.c1);
    Fe t = fe_mul(fe_norm(fe_add(a.c0, a.c1)), fe_add(b.c0, b.c1));
    return {fe_lc3<1, -1, 0>(aa, bb, bb), fe_lc3<1, -1, -1>(t, aa, bb)};
}
// complex squaring: (a0+a1)(a0-a1), 2 a0 a1
BN_FN Fq2A f2_sqr(const Fq2A &a) {
    Fe c0 = fe_mul(fe_add(a.c0, a.c1), fe_sub<1, 4>(a.c0, a.c1));
    Fe c1 = fe_mul(fe_dbl(a.c0), a.c1);
    return {c0, c1};
}
// a may be lazy (lb <= 6, vb <= 56)
BN_FN Fq2A f2_scale(const Fq2A &a, const Fe &s) { return {fe_mul(a.c0, s), fe_mul(a.c1, s)}; }
// reduce(CX * xi * x + CY * y), xi = 9 + i:  xi*x = (9 x0 - x1) + (9 x1 + x0) i.   x, y may be lazy.
template <int CX, int CY>
BN_FN Fq2A f2_lc_xi(const Fq2A &x, const Fq2A &y) {
    return {fe_lc3<9 * CX, -CX, CY>(x.c0, x.c1, y.c0), fe_lc3<9 * CX, CX, CY>(x.c1, x.c0, y.c1)};
}
BN_FN Fq2A f2_mul_xi(const Fq2A &x) { return f2_lc_xi<1, 0>(x, x); }
// lazy variants for values that go straight into a multiplication as the FIRST operand (lb <= 2 there)
BN_FN Fq2A f2_neg_lazy(const Fq2A &a) { return {fe_neg<1, 4>(a.c0), fe_neg<1, 4>(a.c1)}; }
BN_FN Fq2A f2_conj_lazy(const Fq2A &a) { return {a.c0, fe_neg<1, 4>(a.c1)}; }
// multiplication by a table constant (Frobenius / twist coefficients); a may be lazy like f2_mul's first operand
template <class T>
BN_FN Fq2A f2_mul_const(const Fq2A &a, const T &tab) { return f2_mul(a, f2_const((const Fq2A *)nullptr, tab)); }
BN_FN Fq2A f2_inverse(const Fq2A &a) {            // fq2.rs:125-136 with a uniform Fermat inversion of the norm
    Fe n = fe_lc3<1, 1, 0>(fe_sqr(a.c0), fe_sqr(a.c1), a.c1);
    Fe t = fe_inverse(n);
    return {fe_mul(a.c0, t), fe_lc3<-1, 0, 0>(fe_mul(a.c1, t), t, t)};
}
// boundary: 16 u32 words = (c0, c1) in the reference image (fq2.rs:24-29)
BN_FN Fq2A f2_load(const Fq2A *, const uint32_t *w) { return {fe_from_u32x8(w), fe_from_u32x8(w + 8)}; }
BN_FN void f2_store(const Fq2A &a, uint32_t *w) { fe_to_u32x8(a.c0, w); fe_to_u32x8(a.c1, w + 8); }
BN_FN Fe f2_scalar_load(const Fq2A *, const uint32_t *w) { return fe_from_u32x8(w); }

}  // namespace bn254

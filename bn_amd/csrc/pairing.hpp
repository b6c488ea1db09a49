// Optimal-ate pairing on BN254 for the HIP engine: G2 line functions FUSED into the Miller loop (the reference first
// materialises 102 EllCoeffs = 19.6 KB per pairing, src/groups/mod.rs:557-588, then replays them :486-519; here each line
// is consumed the moment it is produced and nothing is stored), then the final exponentiation (src/fields/fq12.rs:41-88).
// Generic over the Fq2 lane mapping.  Entry points used by the kernels:
//     miller_loop<F2>(P affine, Q affine)  ->  Fq12        (groups/mod.rs:486-519 + 557-635)
//     final_exponentiation<F2>(f)          ->  Fq12        (fq12.rs:86-88)
//     g2_to_affine / g1_to_affine                          (groups/mod.rs:113-130; inversion by Fermat, uniform)
#pragma once
#include "tower.hpp"

namespace bn254 {

template <class F2> struct G2Proj { F2 x, y, z; };            // homogeneous projective R of the flipped Miller loop
template <class F2> struct G2Aff { F2 x, y; };
template <class S> struct G1Aff { S x, y; };
template <class F2> struct Line { F2 ell_0, ell_vw, ell_vv; };  // ell_vw / ell_vv are returned LAZY (they only feed f2_scale)

#define F2P ((const F2 *)nullptr)

// groups/mod.rs:612-634.   e = 3b' * z^2 folds the reference's d = 3c, e = b'*d into one constant product.
template <class F2>
BN_COARSE Line<F2> doubling_step(G2Proj<F2> &r) {
    F2 a = f2_scale(f2_mul(r.x, r.y), fe_const(k::TWO_INV));
    F2 b = f2_sqr(r.y), c = f2_sqr(r.z);
    F2 e = f2_mul_const(c, k::G2_3B);
    F2 f3 = f2_add(f2_add(e, e), e);                                     // f = 3e, lazy
    F2 g = f2_scale(f2_add(b, f3), fe_const(k::TWO_INV));                // (b + f)/2
    F2 h = f2_lc3<1, -1, -1>(f2_sqr(f2_lc3<1, 1, 0>(r.y, r.z, r.z)), b, c);
    F2 j = f2_sqr(r.x), e_sq = f2_sqr(e);
    Line<F2> l;
    l.ell_0 = f2_mul_xi(f2_sub<1, 4>(e, b));                             // xi * (e - b)
    l.ell_vw = f2_neg_lazy(h);
    l.ell_vv = f2_add(f2_add(j, j), j);
    r.x = f2_mul(a, f2_lc3<1, -3, 0>(b, e, e));                          // a (b - f)
    r.y = f2_lc3<1, -3, 0>(f2_sqr(g), e_sq, e_sq);                       // g^2 - 3 e^2
    r.z = f2_mul(b, h);
    return l;
}
// groups/mod.rs:592-610
template <class F2>
BN_COARSE Line<F2> addition_step(G2Proj<F2> &r, const G2Aff<F2> &base) {
    F2 d = f2_lc3<1, -1, 0>(r.x, f2_mul(r.z, base.x), r.x);
    F2 e = f2_lc3<1, -1, 0>(r.y, f2_mul(r.z, base.y), r.y);
    F2 f = f2_sqr(d), g = f2_sqr(e);
    F2 h = f2_mul(d, f), i = f2_mul(r.x, f);
    F2 j = f2_lc3<1, -2, 0>(f2_add(f2_mul(r.z, g), h), i, i);
    Line<F2> l;
    l.ell_0 = f2_mul_xi(f2_sub<1, 4>(f2_mul(e, base.x), f2_mul(d, base.y)));
    l.ell_vv = f2_neg_lazy(e);
    l.ell_vw = d;
    F2 ny = f2_lc3<1, -1, 0>(f2_mul(e, f2_lc3<1, -1, 0>(i, j, j)), f2_mul(h, r.y), h);
    r.x = f2_mul(d, j);
    r.y = ny;
    r.z = f2_mul(r.z, h);
    return l;
}
// groups/mod.rs:550-555
template <class F2>
BN_COARSE G2Aff<F2> mul_by_q(const G2Aff<F2> &a) {
    return {f2_mul_const(f2_conj_lazy(a.x), k::TWIST_MUL_BY_Q_X), f2_mul_const(f2_conj_lazy(a.y), k::TWIST_MUL_BY_Q_Y)};
}
// f <- f * line(P)   (groups/mod.rs:502,507,513,516)
template <class F2, class S>
BN_FN Fq12<F2> apply_line(const Fq12<F2> &f, const Line<F2> &l, const G1Aff<S> &p) {
    return f12_mul_by_024(f, l.ell_0, f2_scale(l.ell_vw, p.y), f2_scale(l.ell_vv, p.x));
}

// groups/mod.rs:486-519 fused with :557-588.  Loop bits are compile-time constants (6u+2, top bit skipped): no divergence.
template <class F2, class S>
BN_FN Fq12<F2> miller_loop(const G1Aff<S> &p, const G2Aff<F2> &q) {
    G2Proj<F2> r = {q.x, q.y, f2_one(F2P)};
    Fq12<F2> f = f12_one<F2>();
#pragma unroll 1
    for (int i = 63; i >= 0; --i) {
        Line<F2> l = doubling_step(r);
        f = apply_line(f12_sqr(f), l, p);
        if ((k::ATE_LOOP_LOW64 >> i) & 1) {
            l = addition_step(r, q);
            f = apply_line(f, l, p);
        }
    }
    G2Aff<F2> q1 = mul_by_q(q);
    G2Aff<F2> q2 = mul_by_q(q1);
    q2.y = f2_neg(q2.y);
    Line<F2> l = addition_step(r, q1);
    f = apply_line(f, l, p);
    l = addition_step(r, q2);
    f = apply_line(f, l, p);
    return f;
}

// fq12.rs:229-246 + 97-101: f^u by square-and-multiply (u has 63 bits, top bit consumed by res = f), then conjugate
template <class F2>
BN_COARSE Fq12<F2> exp_by_neg_z(const Fq12<F2> &f) {
    Fq12<F2> res = f;
#pragma unroll 1
    for (int i = 61; i >= 0; --i) {
        res = f12_cyclotomic_sqr(res);
        if ((k::BN_U >> i) & 1) res = f12_mul(f, res);
    }
    return f12_conj(res);
}
// fq12.rs:41-52
template <class F2>
BN_FN Fq12<F2> final_exp_first_chunk(const Fq12<F2> &f) {
    Fq12<F2> b = f12_inverse(f);
    Fq12<F2> c = f12_mul(f12_conj(f), b);
    return f12_mul(f12_frobenius<2>(c), c);
}
// fq12.rs:54-84
template <class F2>
BN_FN Fq12<F2> final_exp_last_chunk(const Fq12<F2> &s) {
    Fq12<F2> a = exp_by_neg_z(s);
    Fq12<F2> b = f12_cyclotomic_sqr(a);
    Fq12<F2> c = f12_cyclotomic_sqr(b);
    Fq12<F2> d = f12_mul(c, b);
    Fq12<F2> e = exp_by_neg_z(d);
    Fq12<F2> f = f12_cyclotomic_sqr(e);
    Fq12<F2> g = exp_by_neg_z(f);
    Fq12<F2> h = f12_conj(d);
    Fq12<F2> i = f12_conj(g);
    Fq12<F2> j = f12_mul(i, e);
    Fq12<F2> kk = f12_mul(j, h);
    Fq12<F2> l = f12_mul(kk, b);
    Fq12<F2> m = f12_mul(kk, e);
    Fq12<F2> n = f12_mul(s, m);
    Fq12<F2> o = f12_frobenius<1>(l);
    Fq12<F2> p = f12_mul(o, n);
    Fq12<F2> q = f12_frobenius<2>(kk);
    Fq12<F2> r = f12_mul(q, p);
    Fq12<F2> t = f12_mul(f12_conj(s), l);
    Fq12<F2> u = f12_frobenius<3>(t);
    return f12_mul(u, r);
}
template <class F2>
BN_FN Fq12<F2> final_exponentiation(const Fq12<F2> &f) { return final_exp_last_chunk(final_exp_first_chunk(f)); }

// groups/mod.rs:113-130 for G2 (z == 1 needs no special case: the general path returns the same canonical values)
template <class F2>
BN_FN G2Aff<F2> g2_to_affine(const F2 &x, const F2 &y, const F2 &z) {
    F2 zi = f2_inverse(z);
    F2 zi2 = f2_sqr(zi);
    return {f2_mul(x, zi2), f2_mul(y, f2_mul(zi2, zi))};
}
BN_FN G1Aff<Fe> g1_to_affine(const Fe &x, const Fe &y, const Fe &z) {
    Fe zi = fe_inverse(z);
    Fe zi2 = fe_sqr(zi);
    return {fe_mul(x, zi2), fe_mul(y, fe_mul(zi2, zi))};
}

#undef F2P
}  // namespace bn254

// One pairing on FOUR lanes (a "quad": two adjacent lane pairs) - the mapping for batches that would leave lane pairs with empty
// SIMDs (round-3 experiment tools/quad_experiment.hip: 1.4-1.5 x faster than the lane-pair mapping up to 16 384 elements per launch,
// 0.8 x from 32 768 on; the host picks it between BN254_OPT_WAVE_PAIRING_MAX and BN254_OPT_QUAD_MAX pairings per call).
//
// An Fq12 f = c0 + c1 w is split over the quad: the LOWER pair (lanes 4k, 4k+1) holds c0, the UPPER pair (4k+2, 4k+3) holds c1, each
// still in the lane-pair Fq2 mapping (fq2.hpp Fq2B: even lane real, odd lane imaginary part).  The pairs talk through
// quad_xchg (DPP quad_perm [2,3,0,1]: every lane reads the same register of the lane two places away).  What is split:
//   f^2 (complex squaring over Fq6, fq12.rs:275-282)   lower: ab = c0 c1,  upper: t = (c0 + c1)(c0 + v c1)     1 Fq6 product instead of 2
//   f * line (fq12.rs:107-176, sparse)                 each pair: own half * (x0 + x2 v^2) + other half * x4    8 Fq2 products instead of 13
//   f * g (fq12.rs:295-307)                            own halves, then the Karatsuba cross term 3 + 3          9 instead of 18
//   Granger-Scott squaring (fq12.rs:178-227)           per Fp4: lower a b, upper (a + b)(a + xi b)              3 instead of 6
// The G2 point arithmetic of the Miller loop is needed by both pairs; its independent same-shape products are dealt out two at a time
// (q_doubling_step: 5 product slots instead of 9, q_addition_step: 7 instead of 13), the linear parts and the inversions run on both
// pairs redundantly.  Per doubling step a pair executes ~20 instead of 34 Fq2-product units, per addition step ~17 instead of 27.
// Every formula below is the lane-pair formula of tower.hpp with its operands picked per role; operand forms (standard / lazy sums)
// are those of the originals, and the host simulation (tests/hostsim/lanequad.hpp) runs this header on a 4-lane value type with
// every bound enforced.  Same field elements, hence the same bytes as the other mappings.
#pragma once
#include "pairing.hpp"

namespace bn254 {

#if !defined(BN_HOSTSIM)
BN_FN Fe quad_xchg(const Fe &x) {                 // the same limb of the OTHER lane pair of the quad
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[i], 0x4E, 0xF, 0xF, true);
    return r;
}
// the lower / the upper pair's value of a register on BOTH pairs: one DPP move per limb where an exchange and two selects used to build
// "mine on my side, the other's on the other side" (quad_perm [0,1,0,1] / [2,3,2,3])
BN_FN Fe quad_lo(const Fe &x) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[i], 0x44, 0xF, 0xF, true);
    return r;
}
BN_FN Fe quad_up(const Fe &x) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[i], 0xEE, 0xF, 0xF, true);
    return r;
}
BN_FN bool quad_is_upper() { return (threadIdx.x & 2u) != 0; }
BN_FN Fe quad_pick(const Fe &lower_choice, const Fe &upper_choice) { return fe_select(quad_is_upper(), lower_choice, upper_choice); }
// an Fq2 of the reference image at w (lower pair) or w + off (upper pair)
BN_FN Fq2B<Fe> quad_load_f2(const Fq2B<Fe> *, const uint32_t *w, int off) { return f2_load((const Fq2B<Fe> *)nullptr, w + (quad_is_upper() ? off : 0)); }
BN_FN void quad_store_f2(const Fq2B<Fe> &a, uint32_t *w, int off) { f2_store(a, w + (quad_is_upper() ? off : 0)); }
#endif

#define F2P ((const F2 *)nullptr)
template <class F2> BN_FN F2 f2_xq(const F2 &a) { return {quad_xchg(a.v)}; }
template <class F2> BN_FN F2 f2_qpick(const F2 &lo, const F2 &up) { return {quad_pick(lo.v, up.v)}; }
template <class F2> BN_FN F2 f2_qlo(const F2 &a) { return {quad_lo(a.v)}; }
template <class F2> BN_FN F2 f2_qup(const F2 &a) { return {quad_up(a.v)}; }
template <class F2> BN_FN Fq6<F2> f6_qlo(const Fq6<F2> &a) { return {f2_qlo(a.c0), f2_qlo(a.c1), f2_qlo(a.c2)}; }
template <class F2> BN_FN Fq6<F2> f6_qup(const Fq6<F2> &a) { return {f2_qup(a.c0), f2_qup(a.c1), f2_qup(a.c2)}; }
template <class F2> BN_FN Fq6<F2> f6_xq(const Fq6<F2> &a) { return {f2_xq(a.c0), f2_xq(a.c1), f2_xq(a.c2)}; }
template <class F2> BN_FN Fq6<F2> f6_qpick(const Fq6<F2> &lo, const Fq6<F2> &up) { return {f2_qpick(lo.c0, up.c0), f2_qpick(lo.c1, up.c1), f2_qpick(lo.c2, up.c2)}; }

// this pair's half of an Fq12: c0 on the lower pair, c1 on the upper pair
template <class F2> struct QFq12 { Fq6<F2> h; };
template <class F2> BN_FN QFq12<F2> q12_one() { return {f6_qpick(f6_one<F2>(), f6_zero<F2>())}; }
template <class F2>
BN_FN QFq12<F2> q12_load(const uint32_t *w) { return {{quad_load_f2(F2P, w, 48), quad_load_f2(F2P, w + 16, 48), quad_load_f2(F2P, w + 32, 48)}}; }
template <class F2>
BN_FN void q12_store(const QFq12<F2> &f, uint32_t *w) { quad_store_f2(f.h.c0, w, 48); quad_store_f2(f.h.c1, w + 16, 48); quad_store_f2(f.h.c2, w + 32, 48); }
// fq12.rs:103-105
template <class F2> BN_FN QFq12<F2> q12_conj(const QFq12<F2> &a) { return {f6_qpick(a.h, f6_neg(a.h))}; }

// fq12.rs:275-282 as tower.hpp f12_sqr<false>: c0' = t - ab - v ab, c1' = 2 ab (carry-propagated sum; every square of the Miller loop
// feeds a sparse product, which takes that form).  ONE Fq6 product per pair.
template <class F2>
BN_COARSE QFq12<F2> q12_sqr(const QFq12<F2> &a) {
    const Fq6<F2> c0 = f6_qlo(a.h), c1 = f6_qup(a.h);
    Fq6<F2> u;                                                    // v c1 + c0
    u.c0 = f2_lc_xi<1, 1>(c1.c2, c0.c0);
    u.c1 = f2_sum_for_mul(c1.c0, c0.c1);
    u.c2 = f2_sum_for_mul(c1.c1, c0.c2);
    const Fq6<F2> x = f6_qpick(c0, f6_add_norm(c0, c1)), y = f6_qpick(c1, u);
    const Fq6<F2> prod = f6_mul(x, y);                             // lower: ab, upper: t
    const Fq6<F2> ab = f6_qlo(prod), t = f6_qup(prod);
    Fq6<F2> lo;
    lo.c0 = f2_lc_xi<-1, 1>(ab.c2, f2_ssub(t.c0, ab.c0));          // t - ab - v ab
    lo.c1 = f2_lc3<1, -1, -1>(t.c1, ab.c1, ab.c0);
    lo.c2 = f2_lc3<1, -1, -1>(t.c2, ab.c2, ab.c1);
    return {f6_qpick(lo, f6_add_norm(ab, ab))};
}

// fq12.rs:107-176: f * (x0 + x2 v^2 + x4 v w).  With A = x0 + x2 v^2 and B = x4 v:  c0' = c0 A + v c1 B,  c1' = c1 A + c0 B, i.e. every
// pair computes  own * A + [v^2 | v] * (other * x4)  (lower | upper): 5 + 3 Fq2 products.
//   own * A = (a0 x0 + xi a1 x2) + (a1 x0 + xi a2 x2) v + (a0 x2 + a2 x0) v^2          (the last by Karatsuba)
//   S = other * x4 coefficient-wise;   v^2 S = (xi s1, xi s2, s0),   v S = (xi s2, s0, s1)
template <class F2>
BN_COARSE QFq12<F2> q12_mul_by_024(const QFq12<F2> &f, const F2 &ell_0, const F2 &ell_vw, const F2 &ell_vv) {
    // by lazy reduction like tower.hpp f12_mul_by_024: every output coefficient is ONE chain of three Fq2 products (fe_mul6) over
    // multipliers prepared once - 9 products, 3 reductions and no recombination per pair instead of 8 products, 8 reductions and 3 fused ones
    const Fq6<F2> o = f6_xq(f.h);
    const F2 &a0 = f.h.c0, &a1 = f.h.c1, &a2 = f.h.c2;
    const auto x0 = f2b_prepare(ell_0), xx2 = f2b_prepare(f2_mul_xi(ell_vv)), x4 = f2b_prepare(ell_vw), xx4 = f2b_prepare(f2_mul_xi(ell_vw));
    decltype(x4) x4m = {quad_pick(xx4.u, x4.u), quad_pick(xx4.v, x4.v)};                       // lower: xi x4, upper: x4
    QFq12<F2> r;
    r.h.c0 = f2b_mul3(a0, x0, a1, xx2, f2_qpick(o.c1, o.c2), xx4);                              // a0 x0 + xi a1 x2 + xi [o1 | o2] x4
    r.h.c1 = f2b_mul3(a1, x0, a2, xx2, f2_qpick(o.c2, o.c0), x4m);                              // a1 x0 + xi a2 x2 + [xi o2 | o0] x4
    const auto x2 = f2b_prepare(ell_vv);
    r.h.c2 = f2b_mul3(a2, x0, a0, x2, f2_qpick(o.c0, o.c1), x4);                                // a2 x0 + a0 x2 + [o0 | o1] x4
    return r;
}
// f <- f * line(P).  The two scalings of the line by P's coordinates (groups/mod.rs:502: ell_vw * P.y, ell_vv * P.x) are the one piece of
// the otherwise redundant point arithmetic that splits for free: the lower pair scales ell_vw, the upper pair ell_vv - ONE Fq product
// per lane instead of two -, and each fetches the other's result.
template <class F2, class S>
BN_FN QFq12<F2> q12_apply_line(const QFq12<F2> &f, const Line<F2> &l, const G1Aff<S> &p) {
    const F2 mine = f2_scale(f2_qpick(l.ell_vw, l.ell_vv), quad_pick(p.y, p.x));
    return q12_mul_by_024(f, l.ell_0, f2_qlo(mine), f2_qup(mine));
}

// fq12.rs:295-307 (Karatsuba over Fq6): `b` hands out THIS pair's half of the multiplier (from the table, or a register copy);
// conj_b multiplies by the conjugate (b0, -b1).  Own halves first (lower a0 b0, upper a1 b1), then the cross term
// (a0 + a1)(b0 + b1) as tower.hpp f6_mul with its six Fq2 products dealt 3 + 3: lower the three s_i t_i, upper the three Karatsuba sums.
template <class F2>
BN_FN QFq12<F2> q12_mul_half(const QFq12<F2> &a, Fq6<F2> b, bool conj_b) {
    if (conj_b) b = f6_qpick(b, f6_neg(b));
    const Fq6<F2> oa = f6_xq(a.h), ob = f6_xq(b);
    const Fq6<F2> mine = f6_mul(a.h, b);                                     // lower: aa, upper: bb
    const Fq6<F2> s = f6_add_norm(a.h, oa), t = f6_add_norm(b, ob);          // a0 + a1, b0 + b1 on both pairs
    const F2 xa = f2_qpick(s.c0, f2_add(s.c1, s.c2)), ya = f2_qpick(t.c0, f2_norm(f2_add(t.c1, t.c2)));
    const F2 xb = f2_qpick(s.c1, f2_add(s.c0, s.c1)), yb = f2_qpick(t.c1, f2_norm(f2_add(t.c0, t.c1)));
    const F2 xc = f2_qpick(s.c2, f2_add(s.c0, s.c2)), yc = f2_qpick(t.c2, f2_norm(f2_add(t.c0, t.c2)));
    const F2 pa = f2_mul(xa, ya), pb = f2_mul(xb, yb), pc = f2_mul(xc, yc);
    const F2 v0 = f2_qlo(pa), v1 = f2_qlo(pb), v2 = f2_qlo(pc);                                 // s_i t_i
    const F2 k12 = f2_qup(pa), k01 = f2_qup(pb), k02 = f2_qup(pc);                              // (s1+s2)(t1+t2), (s0+s1)(t0+t1), (s0+s2)(t0+t2)
    const Fq6<F2> aa = f6_qlo(mine), bb = f6_qup(mine);
    // ONE fused reduction  xi X + Y - Z  per coefficient for both roles (lane-uniform stream: separate reductions - three for the lower
    // pair, three for the cross product and three for the Karatsuba difference of the upper pair - would all run on both pairs):
    //   lower  c0 = aa + v bb:   (xi bb2 + aa0,  aa1 + bb0,  aa2 + bb1)
    //   upper  c1 = s t - aa - bb with f6_mul's recombination of s t inlined:
    //          (xi (k12 - v1 - v2) + v0 - aa0 - bb0,   xi v2 + (k01 - v0 - v1) - aa1 - bb1,   (k02 + v1 - v0 - v2) - aa2 - bb2)
    // Signed lazy operands of up to four terms each: all terms go through the 64-bit chain (f2_lc_xi2w / f2_lc3sw).
    const F2 zero = f2_zero(F2P);
    Fq6<F2> r;
    r.c0 = f2_lc_xi2w<1, 1, -1>(f2_qpick(bb.c2, f2_ssub(f2_ssub(k12, v1), v2)), f2_qpick(aa.c0, f2_ssub(v0, aa.c0)), f2_qpick(zero, bb.c0));
    r.c1 = f2_lc_xi2w<1, 1, -1>(f2_qpick(zero, v2), f2_qpick(aa.c1, f2_ssub(f2_ssub(k01, v0), v1)), f2_qpick(f2_neg_lazy(bb.c0), f2_add(aa.c1, bb.c1)));
    r.c2 = f2_lc3sw<1, -1, 0>(f2_qpick(aa.c2, f2_ssub(f2_ssub(f2_add(k02, v1), v0), v2)), f2_qpick(f2_neg_lazy(bb.c1), f2_add(aa.c2, bb.c2)), zero);
    return {r};
}
template <class F2> BN_OUTER QFq12<F2> q12_mul_o(const QFq12<F2> &a, const QFq12<F2> &b) { return q12_mul_half(a, b.h, false); }

// Granger-Scott (fq12.rs:178-227 as tower.hpp f12_cyclotomic_sqr): the lower pair holds (z0, z4, z3), the upper (z2, z1, z5).  Per Fp4
// square the lower pair takes tmp = a b, the upper m = (a + b)(a + xi b); 3 Fq2 products per pair.
template <class F2>
BN_COARSE QFq12<F2> q12_cyclotomic_sqr(const QFq12<F2> &f) {
    const F2 z0 = f2_qlo(f.h.c0), z4 = f2_qlo(f.h.c1), z3 = f2_qlo(f.h.c2);
    const F2 z2 = f2_qup(f.h.c0), z1 = f2_qup(f.h.c1), z5 = f2_qup(f.h.c2);
    auto fp4 = [&](const F2 &a, const F2 &b, F2 &tmp, F2 &m) {
        const F2 x = f2_qpick(a, f2_add(a, b)), y = f2_qpick(b, f2_lc_xi<1, 1>(b, a));
        const F2 p = f2_mul(x, y);
        tmp = f2_qlo(p); m = f2_qup(p);
    };
    F2 t01, m01, t23, m23, t45, m45;
    fp4(z0, z1, t01, m01); fp4(z2, z3, t23, m23); fp4(z4, z5, t45, m45);
    // ONE fused reduction  -3 xi X + 3 Y - 2 Z  per coefficient serves both roles (the instruction stream is lane-uniform: two
    // role-specific reductions would both be executed by both pairs):
    //   lower  z0' = 3 (m01 - t01 - xi t01) - 2 z0 ...      (X, Y, Z) = (t, m - t, z_even)
    //   upper  z2' = 6 xi t45 + 2 z2                         (X, Y, Z) = (-2 t45, 0, -z2)
    //          z1' = 6 t01 + 2 z1,  z5' = 6 t23 + 2 z5       (X, Y, Z) = (0, 2 t, -z_odd)
    const F2 zero = f2_zero(F2P);
    Fq6<F2> r;
    r.c0 = f2_lc_xi2<-3, 3, -2>(f2_qpick(t01, f2_dbl(f2_neg_lazy(t45))), f2_qpick(f2_ssub(m01, t01), zero), f2_qpick(z0, f2_neg_lazy(z2)));
    r.c1 = f2_lc_xi2<-3, 3, -2>(f2_qpick(t23, zero), f2_qpick(f2_ssub(m23, t23), f2_dbl(t01)), f2_qpick(z4, f2_neg_lazy(z1)));
    r.c2 = f2_lc_xi2<-3, 3, -2>(f2_qpick(t45, zero), f2_qpick(f2_ssub(m45, t45), f2_dbl(t23)), f2_qpick(z3, f2_neg_lazy(z5)));
    return {r};
}

// fq12.rs:284-292: the two Fq6 squares in parallel, d = c0^2 - v c1^2 and its inverse (one Fq inversion) on both pairs, own half * t
template <class F2>
BN_OUTER QFq12<F2> q12_inverse(const QFq12<F2> &a) {
    const Fq6<F2> sq = f6_sqr(a.h);
    const Fq6<F2> s0 = f6_qlo(sq), s1 = f6_qup(sq);
    Fq6<F2> d;
    d.c0 = f2_lc_xi<-1, 1>(s1.c2, s0.c0);
    d.c1 = f2_lc3<1, -1, 0>(s0.c1, s1.c0, s1.c0);
    d.c2 = f2_lc3<1, -1, 0>(s0.c2, s1.c1, s1.c1);
    const Fq6<F2> t = f6_inverse(d);
    const Fq6<F2> r = f6_mul(a.h, t);
    return {f6_qpick(r, f6_neg(r))};
}
// fq12.rs:90-95: the Fq6 map on the own half, the upper pair scales its half by FROB12_C1[P]
template <int P, class F2>
BN_FN QFq12<F2> q12_frobenius(const QFq12<F2> &a) {
    const Fq6<F2> m = f6_frobenius<P>(a.h);
    return {f6_qpick(m, f6_scale(m, f2_const(F2P, k::FROB12_C1[P])))};
}

// ---- the G2 point arithmetic of the Miller loop.  Both pairs need R, the lines and P, so the steps run on both - but wherever a step
// holds TWO independent Fq2 products (or squares) of the same shape, the lower pair computes one, the upper pair the other, and each
// fetches the other's result (36 selects + 9 DPP moves instead of a ~330-instruction product).  Same formulas, operand forms and
// results as pairing.hpp doubling_step<true> / addition_step (groups/mod.rs:612-634, 592-610).
template <class F2>
BN_FN void f2_mul_split(const F2 &a_lo, const F2 &b_lo, const F2 &a_up, const F2 &b_up, F2 &r_lo, F2 &r_up) {
    const F2 m = f2_mul(f2_qpick(a_lo, a_up), f2_qpick(b_lo, b_up));
    r_lo = f2_qlo(m); r_up = f2_qup(m);
}
template <class F2>
BN_FN void f2_sqr_split(const F2 &a_lo, const F2 &a_up, F2 &r_lo, F2 &r_up) {
    const F2 m = f2_sqr(f2_qpick(a_lo, a_up));
    r_lo = f2_qlo(m); r_up = f2_qup(m);
}
// 2 + 3 product slots per pair instead of 3 products + 6 squares
template <class F2>
BN_COARSE Line<F2> q_doubling_step(G2Proj<F2> &r) {
    F2 b, c, j, yz2;
    f2_sqr_split(r.y, r.z, b, c);                                        // b = y^2 | c = z^2
    f2_sqr_split(r.x, f2_sum_for_mul(r.y, r.z), j, yz2);                 // j = x^2 | (y + z)^2
    const F2 a = f2_half(f2_mul(r.x, r.y));
    const F2 e = f2_mul_iso3b(c);                                        // 3 b' t^6 z^2 on the isomorphic curve
    const F2 f3 = f2_add(f2_add(e, e), e);
    const F2 g = f2_half_for_sqr(f2_add(b, f3));
    const F2 h = f2_lc3<1, -1, -1>(yz2, b, c);
    F2 g2, e_sq;
    f2_sqr_split(g, e, g2, e_sq);                                        // g^2 | e^2
    Line<F2> l;
    l.ell_0 = f2_mul_xi(f2_ssub(e, b));
    l.ell_vw = f2_neg_lazy(h);
    l.ell_vv = f2_add(f2_add(j, j), j);
    f2_mul_split(a, f2_lc3<1, -3, 0>(b, e, e), b, h, r.x, r.z);          // x' = a (b - f) | z' = b h
    r.y = f2_lc3<1, -3, 0>(g2, e_sq, e_sq);
    return l;
}
// 6 + 1 product slots per pair instead of 11 products + 2 squares
template <class F2>
BN_COARSE Line<F2> q_addition_step(G2Proj<F2> &r, const G2Aff<F2> &base) {
    F2 zx, zy;
    f2_mul_split(r.z, base.x, r.z, base.y, zx, zy);                      // z bx | z by
    const F2 d = f2_lc3<1, -1, 0>(r.x, zx, r.x), e = f2_lc3<1, -1, 0>(r.y, zy, r.y);
    F2 f, g;
    f2_sqr_split(d, e, f, g);                                            // d^2 | e^2
    F2 h, i;
    f2_mul_split(d, f, r.x, f, h, i);                                    // d f | x f
    F2 zg, ebx;
    f2_mul_split(r.z, g, e, base.x, zg, ebx);                            // z g | e bx
    F2 dby, hy;
    f2_mul_split(d, base.y, h, r.y, dby, hy);                            // d by | h y
    const F2 j = f2_lc3<1, -2, 0>(f2_add(zg, h), i, i);
    Line<F2> l;
    l.ell_0 = f2_mul_xi(f2_ssub(ebx, dby));
    l.ell_vv = f2_neg_lazy(e);
    l.ell_vw = d;
    F2 eij, dj;
    f2_mul_split(e, f2_lc3<1, -1, 0>(i, j, j), d, j, eij, dj);           // e (i - j) | d j
    r.y = f2_lc3<1, -1, 0>(eij, hy, h);
    r.x = dj;
    r.z = f2_mul(r.z, h);
    return l;
}

// The maps as the exponentiation machine calls them: out of line (they are rare), but with the operand BY VALUE in <9 x i32> vectors
// on the GPU - the running value of the machine handed over by reference has its address escape, lives in a stack slot from then on and
// is loaded and stored (27 dwords) in EVERY step (fe.hpp "leaf calling convention"; the same pathology as curve.hpp jac_double_cold)
#if defined(BN_HOSTSIM)
template <int P, class F2> BN_FN QFq12<F2> q12_frobenius_call(const QFq12<F2> &a) { return q12_frobenius<P>(a); }
#else
template <int P, class F2>
BN_OUTER void q12_frobenius_vec(u32x9 c0, u32x9 c1, u32x9 c2, QFq12<F2> *out) {
    QFq12<F2> a;
    a.h.c0.v = bn_unv(c0); a.h.c1.v = bn_unv(c1); a.h.c2.v = bn_unv(c2);
    *out = q12_frobenius<P>(a);
}
template <int P, class F2> BN_FN QFq12<F2> q12_frobenius_call(const QFq12<F2> &a) {
    QFq12<F2> t;
    q12_frobenius_vec<P, F2>(bn_tov(a.h.c0.v), bn_tov(a.h.c1.v), bn_tov(a.h.c2.v), &t);
    return t;
}
#endif

// ---- Miller loop: miller_loop_sched<true> (NAF schedule on the isomorphic curve) with f split over the quad; R, the point being
// added and P live in `st` (LDS in the kernel) on BOTH pairs
template <class F2, class S, class Store>
BN_FN QFq12<F2> q_miller_loop_naf(const G1Aff<S> &p_in, const G2Aff<F2> &q_in, Store &st) {
    {
        const S t2 = f2_scalar_const(F2P, k::ISO_T2), t3 = f2_scalar_const(F2P, k::ISO_T3);
        G1Aff<S> p = {fe_mul(p_in.x, t2), fe_mul(p_in.y, t3)};
        G2Aff<F2> q = {f2_scale(q_in.x, t2), f2_scale(q_in.y, t3)};
        G2Proj<F2> r0 = {q.x, q.y, f2_one(F2P)};
        st.put_r(r0);
        st.put_base(q);
        st.put_p(p);
    }
    QFq12<F2> f = q12_one<F2>();
    constexpr int ND = k::ATE_NAF_LEN - 1;
#pragma unroll 1
    for (int j = 0; j < ND + 2; ++j) {
        const bool tail = j >= ND;
        const int digit = tail ? 1 : k::ATE_NAF[ND - 1 - j];
        if (j == ND) st.put_base(mul_by_q(st.get_base()));                          // pi(Q)      groups/mod.rs:578
        if (j == ND + 1) { G2Aff<F2> b2 = mul_by_q(st.get_base()); b2.y = f2_neg(b2.y); st.put_base(b2); }   // -pi^2(Q)   :579
#pragma unroll 1
        for (int pass = tail ? 1 : 0; pass < (digit != 0 ? 2 : 1); ++pass) {
            Line<F2> l;
            if (pass == 0) {
                if (j != 0) f = q12_sqr(f);
                BN_COMPILER_FENCE();
                G2Proj<F2> r = st.get_r();
                l = q_doubling_step(r);
                st.put_r(r);
            } else {
                G2Proj<F2> r = st.get_r();
                G2Aff<F2> b = st.get_base();
                if (digit < 0) b.y = f2_neg(b.y);
                l = q_addition_step(r, b);
                st.put_r(r);
            }
            f = q12_apply_line(f, l, st.get_p());
        }
    }
    return f;
}

// ---- final exponentiation: fq12.rs:41-52 (easy part) and the engine's program of the hard part (pairing.hpp fe_step / k::FE_PROG),
// every operation in its quad form.  `tbl`: put(slot, half) / get(slot) of THIS pair's Fq6 halves.
template <class F2> struct QuadTableVars {
    Fq6<F2> s_[k::EXP_SLOTS];
    BN_FN void put(int i, const Fq6<F2> &v) { s_[i] = v; }
    BN_FN Fq6<F2> get(int i) const { return s_[i]; }
};
template <class F2, class Tbl>
BN_FN void q_fe_step(QFq12<F2> &res, const int w, Tbl &tbl) {
    const int get = (w >> 10) & 15, mul = (w >> 1) & 15, put = (w >> 6) & 15, post = (w >> 14) & 7;
    if (get) res = QFq12<F2>{tbl.get(get - 1)};
    if (w & 1) res = q12_cyclotomic_sqr(res);
    if (mul) res = q12_mul_half(res, tbl.get(mul - 1), ((w >> 5) & 1) != 0);
    if (post == 1) {
        res = q12_conj(res);
    } else if (post) {
        res = post == 2 ? q12_frobenius_call<1>(res) : post == 3 ? q12_frobenius_call<2>(res) : q12_frobenius_call<3>(res);
    }
    if (put) tbl.put(put - 1, res.h);
}
template <class F2, class Tbl>
BN_FN QFq12<F2> q_final_exponentiation(const QFq12<F2> &f, Tbl &tbl) {
    const QFq12<F2> b = q12_inverse(f);
    const QFq12<F2> c = q12_mul_o(q12_conj(f), b);
    // (a separate variable for the loop: the out-of-line product writes its result through a pointer, and a loop-carried value whose
    //  address has escaped stays in memory for the whole program)
    const QFq12<F2> start = q12_mul_o(q12_frobenius_call<2>(c), c);
    QFq12<F2> res = {start.h};
#pragma unroll 1
    for (int i = 0; i < k::FE_STEPS; ++i) q_fe_step(res, k::FE_PROG[i], tbl);
    return res;
}

#undef F2P
}  // namespace bn254

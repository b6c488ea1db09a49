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
// ISO = true: the step on the isomorphic curve y^2 = x^3 + b' t^6 (t^6 = 82/3, bn254_constants.hpp ISO_T2/ISO_T3), whose constant
// 3 b' t^6 = 27 - 3i costs a two-term linear combination instead of a product.  Only the pairing kernels use it (with P and Q
// mapped by (x, y) -> (t^2 x, t^3 y) first): the reduced pairing is invariant under the isomorphism, the Miller value is not.
template <bool ISO = false, class F2>
BN_COARSE Line<F2> doubling_step(G2Proj<F2> &r) {
    F2 a = f2_half(f2_mul(r.x, r.y));                                    // x y / 2 by a shift, not a product (fe_half; profiles/r02h_ab_half_vs_scale.txt)
    F2 b = f2_sqr(r.y), c = f2_sqr(r.z);
    F2 e;
    if constexpr (ISO) e = f2_mul_iso3b(c);
    else e = f2_mul_const(c, k::G2_3B);
    F2 f3 = f2_add(f2_add(e, e), e);                                     // f = 3e, lazy
    F2 g = f2_half_for_sqr(f2_add(b, f3));                               // (b + f)/2
    F2 h = f2_lc3<1, -1, -1>(f2_sqr(f2_sum_for_mul(r.y, r.z)), b, c);
    F2 j = f2_sqr(r.x), e_sq = f2_sqr(e);
    Line<F2> l;
    l.ell_0 = f2_mul_xi(f2_ssub(e, b));                             // xi * (e - b)
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
    l.ell_0 = f2_mul_xi(f2_ssub(f2_mul(e, base.x), f2_mul(d, base.y)));
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
BN_OUTER G2Aff<F2> mul_by_q(const G2Aff<F2> &a) {
    return {f2_mul_const(f2_conj_lazy(a.x), k::TWIST_MUL_BY_Q_X), f2_mul_const(f2_conj_lazy(a.y), k::TWIST_MUL_BY_Q_Y)};
}
// f <- f * line(P)   (groups/mod.rs:502,507,513,516)
template <class F2, class S>
BN_FN Fq12<F2> apply_line(const Fq12<F2> &f, const Line<F2> &l, const G1Aff<S> &p) {
    return f12_mul_by_024(f, l.ell_0, f2_scale(l.ell_vw, p.y), f2_scale(l.ell_vv, p.x));
}

// Where the slowly changing state of the Miller loop (the running point R, the point being added, the affine P) lives
// BETWEEN steps.  Default: ordinary variables.  The lane-pair kernel parks them in LDS (pairing kernels keep f in VGPRs and
// would otherwise spill exactly these values to private memory = HBM traffic).
template <class F2, class S>
struct MillerStateVars {
    G2Proj<F2> r_;
    G2Aff<F2> base_;
    G1Aff<S> p_;
    BN_FN void put_r(const G2Proj<F2> &v) { r_ = v; }
    BN_FN G2Proj<F2> get_r() const { return r_; }
    BN_FN void put_base(const G2Aff<F2> &v) { base_ = v; }
    BN_FN G2Aff<F2> get_base() const { return base_; }
    BN_FN void put_p(const G1Aff<S> &v) { p_ = v; }
    BN_FN G1Aff<S> get_p() const { return p_; }
};

// groups/mod.rs:486-519 fused with :557-588.  The schedule (6u+2 with the top bit skipped: 64 doublings, an addition of Q
// after every set bit, then the additions of pi(Q) and -pi^2(Q)) is a compile-time constant, so every branch below is
// wave-uniform.  Written as ONE loop of steps x up to 2 passes so that each of f^2, the two line functions and the
// sparse multiplication exists exactly once in the instruction stream.
//   NAF = false: the reference's schedule; the returned Miller value equals the reference's miller_loop limb for limb.
//   NAF = true : 6u+2 in non-adjacent form (65 doublings, 21 additions of +-Q instead of 64 + 36).  The Miller value differs
//                from the reference's by a factor that the final exponentiation kills (different line scalings / omitted
//                verticals, all in proper subfields), so pairing() - the only observable of this path - is bit-identical.
template <bool NAF, class F2, class S, class Store>
BN_FN Fq12<F2> miller_loop_sched(const G1Aff<S> &p_in, const G2Aff<F2> &q_in, Store &st) {
    {
        G1Aff<S> p = p_in;
        G2Aff<F2> q = q_in;
        if constexpr (NAF) {             // onto the isomorphic curve: (x, y) -> (t^2 x, t^3 y) for both points (profiles/r02i_ab_isomorphic_curve.txt)
            const S t2 = f2_scalar_const(F2P, k::ISO_T2), t3 = f2_scalar_const(F2P, k::ISO_T3);
            p = {fe_mul(p.x, t2), fe_mul(p.y, t3)};
            q = {f2_scale(q.x, t2), f2_scale(q.y, t3)};
        }
        G2Proj<F2> r0 = {q.x, q.y, f2_one(F2P)};
        st.put_r(r0);
        st.put_base(q);
        st.put_p(p);
    }
    Fq12<F2> f = f12_one<F2>();
    constexpr int ND = NAF ? k::ATE_NAF_LEN - 1 : 64;                                // doublings
#pragma unroll 1
    for (int j = 0; j < ND + 2; ++j) {
        const bool tail = j >= ND;
        int digit = 1;
        if (!tail) {
            if (NAF) digit = k::ATE_NAF[ND - 1 - j];
            else digit = (int)((k::ATE_LOOP_LOW64 >> (63 - (j & 63))) & 1);
        }
        if (j == ND) st.put_base(mul_by_q(st.get_base()));                          // pi(Q)      groups/mod.rs:578
        if (j == ND + 1) { G2Aff<F2> b2 = mul_by_q(st.get_base()); b2.y = f2_neg(b2.y); st.put_base(b2); }   // -pi^2(Q)   :579
#pragma unroll 1
        for (int pass = tail ? 1 : 0; pass < (digit != 0 ? 2 : 1); ++pass) {
            BN_MILLER_HOOK(2 * j + pass, 2 * (ND + 2));
            Line<F2> l;
            if (pass == 0) {
                if (j != 0) f = f12_sqr(f);                                         // f == 1 in the first step
                BN_COMPILER_FENCE();                                                // R is fetched AFTER the squaring
                G2Proj<F2> r = st.get_r();
                l = doubling_step<NAF>(r);
                st.put_r(r);
            } else {
                G2Proj<F2> r = st.get_r();
                G2Aff<F2> b = st.get_base();
                if (NAF && digit < 0) b.y = f2_neg(b.y);
                l = addition_step(r, b);
                st.put_r(r);
            }
            // (replacing this product by an assignment in the very first step, where f == 1, was measured 10 % SLOWER: the extra
            // control flow costs the register allocation of the whole loop more than the 0.5 % of multiplications it saves)
            f = apply_line(f, l, st.get_p());
        }
    }
    return f;
}
// (Multiplying the two lines of a step that adds +-Q together BEFORE they meet f - 6 + 17 instead of 13 + 13 Fq2 products on 21 of the 88
// steps, 2.4 % fewer multiply-adds - was implemented, bit-identical and measured 2 % SLOWER: f has to wait in registers through the addition
// step and the line product, and the allocator spills ~100 VGPRs of it.  profiles/r02x_ab_merge_lines.txt; the code left with round 5.)
// The multi-pairing's Miller loop with a SHARED accumulator: prod_i f_i = the f of  f <- f^2 * prod_i l_i(P_i)  (a product of Miller
// values only ever meets one final exponentiation: shootout/main.rs:11-16), so M pairs on one lane pair pay ONE f^2 per doubling
// step - 12 of the 33 Fq2 products of a step and pair - instead of M.  The price: the running points, the points being added and the
// affine P_i of all M pairs do not fit in LDS next to f; `st` keeps them in global memory (st.get_r(i) ...).  NAF schedule on the
// isomorphic curve, as miller_loop_sched<true>.  `st.is_inf(i)`: pair i contributes 1 (groups/mod.rs:766) - its lines are replaced
// by the constant 1, so f is untouched.  The prologue (st.put_* for every i) is the caller's.
template <int M, class F2, class S, class Store>
BN_FN Fq12<F2> miller_loop_shared(Store &st) {
    Fq12<F2> f = f12_one<F2>();
    constexpr int ND = k::ATE_NAF_LEN - 1;
#pragma unroll 1
    for (int j = 0; j < ND + 2; ++j) {
        const bool tail = j >= ND;
        const int digit = tail ? 1 : k::ATE_NAF[ND - 1 - j];
        if (tail) {
#pragma unroll 1
            for (int i = 0; i < M; ++i) {                                            // pi(Q_i), then -pi^2(Q_i)   groups/mod.rs:578-579
                G2Aff<F2> b = mul_by_q(st.get_base(i));
                if (j == ND + 1) b.y = f2_neg(b.y);
                st.put_base(i, b);
            }
        }
#pragma unroll 1
        for (int pass = tail ? 1 : 0; pass < (digit != 0 ? 2 : 1); ++pass) {
            BN_MILLER_HOOK(2 * j + pass, 2 * (ND + 2));
            if (pass == 0 && j != 0) f = f12_sqr(f);                                // ONE squaring for all M pairs
#pragma unroll 1
            for (int i = 0; i < M; ++i) {
                BN_COMPILER_FENCE();
                G2Proj<F2> r = st.get_r(i);
                Line<F2> l;
                if (pass == 0) {
                    l = doubling_step<true>(r);
                } else {
                    G2Aff<F2> b = st.get_base(i);
                    if (digit < 0) b.y = f2_neg(b.y);
                    l = addition_step(r, b);
                }
                st.put_r(i, r);
                const G1Aff<S> p = st.get_p(i);
                F2 x0 = l.ell_0, x4 = f2_scale(l.ell_vw, p.y), x2 = f2_scale(l.ell_vv, p.x);
                const bool inf = st.is_inf(i);                                      // an infinite pair multiplies by the line 1 + 0 + 0
                x0 = f2_select(inf, x0, f2_one(F2P)); x4 = f2_select(inf, x4, f2_zero(F2P)); x2 = f2_select(inf, x2, f2_zero(F2P));
                f = f12_mul_by_024(f, x0, x4, x2);
            }
        }
    }
    return f;
}
template <class F2, class S, class Store>
BN_FN Fq12<F2> miller_loop(const G1Aff<S> &p, const G2Aff<F2> &q, Store &st) { return miller_loop_sched<false>(p, q, st); }
template <class F2, class S>
BN_FN Fq12<F2> miller_loop(const G1Aff<S> &p, const G2Aff<F2> &q) {
    MillerStateVars<F2, S> st;
    return miller_loop(p, q, st);
}

// ---- prepared-G2 mode (the reference's internal G2Precomp: groups/mod.rs:472-483, precompute :557-588, miller_loop :486-519)
// precompute_lines: the 102 line coefficients of an affine Q, in schedule order, handed to `sink(index, line)`.
template <class F2, class Sink>
BN_FN void precompute_lines(const G2Aff<F2> &q, Sink &sink) {
    G2Proj<F2> r = {q.x, q.y, f2_one(F2P)};
    G2Aff<F2> base = q;
    int idx = 0;
#pragma unroll 1
    for (int j = 0; j < 66; ++j) {
        const bool tail = j >= 64;
        const bool bit = tail ? true : (((k::ATE_LOOP_LOW64 >> (63 - (j & 63))) & 1) != 0);
        if (j == 64) base = mul_by_q(base);
        if (j == 65) { base = mul_by_q(base); base.y = f2_neg(base.y); }
#pragma unroll 1
        for (int pass = tail ? 1 : 0; pass < (bit ? 2 : 1); ++pass) {
            BN_MILLER_HOOK(2 * j + pass, 2 * 66);
            Line<F2> l = pass == 0 ? doubling_step<false>(r) : addition_step(r, base);
            sink(idx++, l);
        }
    }
}
// miller_loop over stored coefficients: `source(index)` returns line `index` (all three members in standard form)
// `pstore.get_p()` hands out the affine P (the lane-pair kernel parks it in LDS between uses, like the fused loop)
template <class F2, class PStore, class Source>
BN_FN Fq12<F2> miller_loop_prepared(const PStore &pstore, Source &source) {
    Fq12<F2> f = f12_one<F2>();
    int idx = 0;
#pragma unroll 1
    for (int j = 0; j < 66; ++j) {
        const bool tail = j >= 64;
        const bool bit = tail ? true : (((k::ATE_LOOP_LOW64 >> (63 - (j & 63))) & 1) != 0);
#pragma unroll 1
        for (int pass = tail ? 1 : 0; pass < (bit ? 2 : 1); ++pass) {
            BN_MILLER_HOOK(2 * j + pass, 2 * 66);
            if (pass == 0) f = f12_sqr(f);
            BN_COMPILER_FENCE();                                                    // the coefficients are fetched AFTER the squaring
            Line<F2> l = source(idx++);
            f = apply_line(f, l, pstore.get_p());
        }
    }
    return f;
}

// ---- NATIVE prepared-G2 mode: the device's own counterpart of G2Precomp (groups/mod.rs:472-483) for workloads that pair many P against
// the same Q (or re-use a set of Q).  Not the reference image (that is precompute_lines / miller_loop_prepared above, kept for the
// reference's known answers) but what the lane-pair kernels consume without any per-pairing conversion:
//   * the schedule is this engine's - 6u+2 in non-adjacent form on the isomorphic curve (miller_loop_sched<true>): NATIVE_LINES = 88 lines
//     (65 doublings, 21 additions of +-Q, pi(Q), -pi^2(Q)) instead of the reference's 102;
//   * every line is divided by ell_vw t^3 (t: the isomorphism's parameter, folded in here so that P stays on the original curve):
//         A = ell_0 / (ell_vw t^3),   B = ell_vv t^2 / (ell_vw t^3)          - ONE inversion per table (Montgomery's trick over the 88 lines)
//     and stored in the radix-2^261 limbs as the multiplier operands of the lane-pair product: this lane's component of A, and B, xi B each as
//     (u, v) = (own, partner's with the even lane's negated) - canonical field elements, so a table is a function of Q alone;
//   * per pairing the kernel computes sigma = 1 / x_P and tau = y_P / x_P behind ONE inversion (of X Z, for a Jacobian P) and multiplies f by
//     A sigma + tau v w + B v^2 (tower.hpp f12_mul_by_line_native) - equal to the reference's line up to a factor in Fq2.  (x_P != 0 on every
//     point of G1: 3 is a non-residue mod q, so y^2 = x^3 + 3 has no point with x = 0.)
// The Miller value differs from the reference's by factors the final exponentiation kills; pairing() is bit-identical.
constexpr int native_line_count() {
    int n = 2;
    for (int j = 0; j < k::ATE_NAF_LEN - 1; ++j) n += 1 + (k::ATE_NAF[j] != 0 ? 1 : 0);
    return n;
}
constexpr int NATIVE_LINES = native_line_count();
static_assert(NATIVE_LINES == 88, "line count of the NAF schedule");
template <class T> BN_FN T fe_canon(const T &x) { return fe_canonical(fe_mul(x, lane_bcast((const T *)nullptr, fe_one()))); }
// (u, v) of a multiplier as canonical field elements (f2b_prepare's v is a lazy negation: fine for a value that lives for one line product,
// not for a table that should depend on Q alone)
template <class T>
BN_FN Fq2BPrep<T> f2b_prepare_canonical(const Fq2B<T> &b) {
    const T own = fe_canon(b.v), pb = lane_partner(own);
    return {lane_pick(own, pb), lane_pick(fe_canon(fe_neg<1, 4>(pb)), own)};
}
// `st`: put_raw(i, ell_0, d, c, prefix) / get_raw(i, ell_0, d, c) / get_prefix(i) park four Fq2 per line between the passes (the kernel uses
// the line's own table record), put_final(i, a_own, b, xi_b) writes the record
template <class F2, class Store>
BN_FN void precompute_native(const G2Aff<F2> &q_in, Store &st) {
    typedef typename F2::Scalar S;
    constexpr int ND = k::ATE_NAF_LEN - 1;
    const S t2 = f2_scalar_const(F2P, k::ISO_T2), t3 = f2_scalar_const(F2P, k::ISO_T3);
    G2Aff<F2> base = {f2_scale(q_in.x, t2), f2_scale(q_in.y, t3)};
    G2Proj<F2> r = {base.x, base.y, f2_one(F2P)};
    F2 pref = f2_one(F2P);
    int idx = 0;
#pragma unroll 1
    for (int j = 0; j < ND + 2; ++j) {
        const bool tail = j >= ND;
        const int digit = tail ? 1 : k::ATE_NAF[ND - 1 - j];
        if (j == ND) base = mul_by_q(base);                                                 // pi(Q)      groups/mod.rs:578
        if (j == ND + 1) { base = mul_by_q(base); base.y = f2_neg(base.y); }                // -pi^2(Q)   :579
#pragma unroll 1
        for (int pass = tail ? 1 : 0; pass < (digit != 0 ? 2 : 1); ++pass) {
            Line<F2> l;
            if (pass == 0) {
                l = doubling_step<true>(r);
            } else {
                G2Aff<F2> b = base;
                if (digit < 0) b.y = f2_neg(b.y);
                l = addition_step(r, b);
            }
            const F2 d = f2_scale(l.ell_vw, t3), c = f2_scale(l.ell_vv, t2);
            pref = f2_mul(pref, d);
            st.put_raw(idx, l.ell_0, d, c, pref);
            ++idx;
        }
    }
    F2 inv = f2_inverse(pref);                                                              // 1 / (d_0 ... d_87)
#pragma unroll 1
    for (idx = NATIVE_LINES - 1; idx >= 0; --idx) {
        F2 e0, d, c;
        st.get_raw(idx, e0, d, c);
        F2 dinv = inv;
        if (idx > 0) { dinv = f2_mul(inv, st.get_prefix(idx - 1)); inv = f2_mul(inv, d); }
        const F2 a = f2_mul(e0, dinv), b = f2_mul(c, dinv);
        st.put_final(idx, fe_canon(a.v), f2b_prepare_canonical(b), f2b_prepare_canonical(f2_mul_xi(b)));
    }
}
// what a pairing needs of P in native mode, from its Jacobian coordinates (x = X / Z^2, y = Y / Z^3): ONE inversion
template <class T> struct PNative { T sigma, tau, tau9, taum; };
template <class T> BN_FN T p_native_tau9(const T &tau) { return fe_lc3<9, 0, 0>(tau, tau, tau); }
template <class T> BN_FN T p_native_taum(const T &tau) { return lane_pick(fe_lc3<-1, 0, 0>(tau, tau, tau), tau); }
template <class T>
BN_FN PNative<T> p_native(const T &x, const T &y, const T &z) {
    const T inv = fe_inverse(fe_mul(x, z));                          // 1 / (X Z)
    PNative<T> p;
    p.sigma = fe_mul(fe_mul(fe_sqr(z), z), inv);                     // Z^2 / X = 1 / x_P
    p.tau = fe_mul(y, inv);                                          // Y / (Z X) = y_P / x_P
    p.tau9 = p_native_tau9(p.tau);                                   // xi z tau = own (9 tau) + partner (-+ tau)
    p.taum = p_native_taum(p.tau);
    return p;
}
// `src.set_line(i)` selects line i; the rest of its interface is f12_mul_by_line_native's
template <class F2, class Src>
BN_FN Fq12<F2> miller_loop_native(Src &src) {
    Fq12<F2> f = f12_one<F2>();
    constexpr int ND = k::ATE_NAF_LEN - 1;
    int idx = 0;
#pragma unroll 1
    for (int j = 0; j < ND + 2; ++j) {
        const bool tail = j >= ND;
        const int digit = tail ? 1 : k::ATE_NAF[ND - 1 - j];
#pragma unroll 1
        for (int pass = tail ? 1 : 0; pass < (digit != 0 ? 2 : 1); ++pass) {
            BN_MILLER_HOOK(2 * j + pass, 2 * (ND + 2));
            if (pass == 0 && j != 0) f = f12_sqr(f);                                    // f == 1 in the first step
            BN_COMPILER_FENCE();                                                        // the line is fetched AFTER the squaring
            src.set_line(idx++);
            f = f12_mul_by_line_native(f, src);
        }
    }
    return f;
}

// The multi-pairing over prepared points: prod_i e(p_i, Q_i) needs ONE final exponentiation, so M pairs on a lane pair share the accumulator
// f (one f^2 per step instead of M: the idea of miller_loop_shared) - and over native tables a pair has NO per-step point state at all: what it
// keeps is sigma and tau (LDS; 9 tau and -+tau are re-derived per line, two fused reductions against a line's ~3.8 k instructions, because 36
// dwords x 4 pairs per lane do not fit the LDS of two waves per SIMD) and the number of its table column.  A pair with a point at infinity
// contributes the factor 1 (groups/mod.rs:766) WITHOUT a select in the loop: it reads the IDENTITY column every native table ends with
// (A = 1, B = xi B = 0: bn254_kernels_b.hip) with sigma = 1, tau = 0, and its "line" A sigma + tau v w + B v^2 is exactly one.
//   src.set_line(line, i): select line `line` of pair i;  the rest of src's interface is f12_mul_by_line_native's
template <class T> BN_FN void p_native_identity(T &sigma, T &tau) {
    sigma = lane_bcast((const T *)nullptr, fe_one());
    tau = lane_bcast((const T *)nullptr, fe_zero());
}
template <int M, class F2, class Src>
BN_FN Fq12<F2> miller_loop_native_shared(Src &src) {
    Fq12<F2> f = f12_one<F2>();
    constexpr int ND = k::ATE_NAF_LEN - 1;
    int idx = 0;
#pragma unroll 1
    for (int j = 0; j < ND + 2; ++j) {
        const bool tail = j >= ND;
        const int digit = tail ? 1 : k::ATE_NAF[ND - 1 - j];
#pragma unroll 1
        for (int pass = tail ? 1 : 0; pass < (digit != 0 ? 2 : 1); ++pass) {
            BN_MILLER_HOOK(2 * j + pass, 2 * (ND + 2));
            if (pass == 0 && j != 0) f = f12_sqr(f);                                    // ONE squaring for all M pairs
#pragma unroll 1
            for (int i = 0; i < M; ++i) {
                BN_COMPILER_FENCE();
                src.set_line(idx, i);
                f = f12_mul_by_line_native(f, src);
            }
            ++idx;
        }
    }
    return f;
}

// Table of the exponentiation machine below: EXP_SLOTS Fq12 values per pairing.  Default: ordinary variables (host
// simulation, one-lane mapping); the lane-pair kernel keeps it in global memory (54 dwords per lane and slot, coalesced), which
// also takes the multiplier of the loop out of the register file.
template <class F2>
struct ExpTableVars {
    Fq12<F2> s_[k::EXP_SLOTS];
    BN_FN void put(int i, const Fq12<F2> &v) { s_[i] = v; }
    BN_FN Fq6<F2> c0(int i) const { return s_[i].c0; }
    BN_FN Fq6<F2> c1(int i) const { return s_[i].c1; }
};
template <class F2, class Tbl>
struct Fq12Slot {
    const Tbl &t;
    int i;
    BN_FN Fq6<F2> c0() const { return t.c0(i); }
    BN_FN Fq6<F2> c1() const { return t.c1(i); }
};

// One step of the exponentiation machine on the cyclotomic subgroup.  Control word (tools/gen_device_constants.py, which also
// executes every program symbolically): load res from a slot | cyclotomic squaring | multiply by a slot (or its conjugate) |
// conjugate / Frobenius map | store res.  Every branch is wave-uniform; each step body exists once in the instruction stream.
template <class F2, class Tbl>
BN_FN void fe_step(Fq12<F2> &res, const int w, Tbl &tbl) {
    const int get = (w >> 10) & 15, mul = (w >> 1) & 15, put = (w >> 6) & 15, post = (w >> 14) & 7;
    if (get) res = Fq12<F2>{tbl.c0(get - 1), tbl.c1(get - 1)};
    if (w & 1) res = f12_cyclotomic_sqr(res);
    if (mul) res = f12_mul_src(res, Fq12Slot<F2, Tbl>{tbl, mul - 1}, ((w >> 5) & 1) != 0);
    if (post == 1) {
        res.c1 = f6_neg(res.c1);                                                    // fq12.rs:103-105
    } else if (post) {
        // the maps are out of line; they get a COPY (a reference to the loop-carried value would pin it to a stack slot)
        const Fq12<F2> cur = res;
        res = post == 2 ? f12_frobenius<1>(cur) : post == 3 ? f12_frobenius<2>(cur) : f12_frobenius<3>(cur);
    }
    if (put) tbl.put(put - 1, res);
}

// fq12.rs:229-246 + 97-101: f^u, then conjugate.  The reference walks the 63 bits of u (62 cyclotomic squarings, 27
// multiplications).  On the cyclotomic subgroup f^-1 = conj(f) is free, so the same group element is reached through the signed
// digits +-17, +-35 of u (62 squarings, 13 multiplications including the table; tools/gen_device_constants.py); the value - and
// therefore every output byte - is identical.  Only valid for f in the cyclotomic subgroup.
template <class F2, class Tbl>
BN_OUTER Fq12<F2> exp_by_neg_z(const Fq12<F2> &f, Tbl &tbl) {
    Fq12<F2> res = f;
#pragma unroll 1
    for (int s = 0; s < k::EXP_STEPS; ++s) {
        BN_EXP_HOOK(s, k::EXP_STEPS);
        fe_step(res, k::EXP_SCHED[s], tbl);
    }
    return res;
}
template <class F2>
BN_FN Fq12<F2> exp_by_neg_z(const Fq12<F2> &f) {
    ExpTableVars<F2> tbl;
    return exp_by_neg_z(f, tbl);
}
// the reference's own schedule (plain binary expansion of u); kept for inputs OFF the cyclotomic subgroup - the known-answer
// test of fields/mod.rs:171-201 feeds exp_by_neg_z such an element, where conj(f) != f^-1
template <class F2>
BN_OUTER Fq12<F2> exp_by_neg_z_reference_schedule(const Fq12<F2> &f) {
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
    Fq12<F2> c = f12_mul_o(f12_conj(f), b);
    return f12_mul_o(f12_frobenius<2>(c), c);
}
// fq12.rs:54-84: the whole hard part (three exponentiations by u and the products, squarings, conjugations and Frobenius maps
// a .. u between them) as ONE program of the machine above (k::FE_PROG, checked symbolically against the reference's sequence):
// the running value stays in registers, the nine values that must be kept live in the table, nothing goes through private memory.
template <class F2, class Tbl>
BN_FN Fq12<F2> final_exp_last_chunk(const Fq12<F2> &s, Tbl &tbl) {
    Fq12<F2> res = s;
#pragma unroll 1
    for (int i = 0; i < k::FE_STEPS; ++i) {
        BN_EXP_HOOK(i, k::FE_STEPS);
        fe_step(res, k::FE_PROG[i], tbl);
    }
    return res;
}
template <class F2, class Tbl>
BN_FN Fq12<F2> final_exponentiation(const Fq12<F2> &f, Tbl &tbl) { return final_exp_last_chunk(final_exp_first_chunk(f), tbl); }
template <class F2>
BN_FN Fq12<F2> final_exponentiation(const Fq12<F2> &f) {
    ExpTableVars<F2> tbl;
    return final_exponentiation(f, tbl);
}

// Gt::pow (lib.rs:171 -> fields/mod.rs:35-46: 256 x { res = res^2; if bit { res = a * res } } on the scalar taken out of Montgomery
// form).  The power is a unique field element, so any addition chain returns the reference's bytes: fixed 4-bit windows, MSB first -
// 256 squarings + 64 table products + a 14-operation table.  General Fq12 squarings (not cyclotomic ones): correct for ANY Fq12.
// `tbl`: 16 Fq12 entries with put(e, v) / c0(e) / c1(e) (per-lane global memory in the kernel, plain variables in the host simulation).
template <class F2, class Tbl>
BN_FN Fq12<F2> gt_pow_windowed(const Fq12<F2> &base, const uint32_t *k_raw, Tbl &tbl) {
    tbl.put(0, f12_one<F2>());
    tbl.put(1, base);
#pragma unroll 1
    for (int e = 2; e < 16; ++e) {                 // a^e = (a^(e/2))^2 for even e, a^(e-1) * a for odd e
        Fq12<F2> v;
        if ((e & 1) == 0) v = f12_sqr<true>(Fq12<F2>{tbl.c0(e >> 1), tbl.c1(e >> 1)});
        else v = f12_mul_src(Fq12<F2>{tbl.c0(e - 1), tbl.c1(e - 1)}, Fq12Slot<F2, Tbl>{tbl, 1}, false);
        tbl.put(e, v);
    }
    Fq12<F2> res = f12_one<F2>();
#pragma unroll 1
    for (int w = 63; w >= 0; --w) {
        BN_EXP_HOOK(63 - w, 64);
#pragma unroll 1
        for (int d = 0; d < 4; ++d) res = f12_sqr<true>(res);
        const int digit = (int)((k_raw[w >> 3] >> ((w & 7) * 4)) & 15u);          // per lane pair: both lanes hold the same scalar
        res = f12_mul_src(res, Fq12Slot<F2, Tbl>{tbl, digit}, false);
    }
    return res;
}
// Gt::pow on the CYCLOTOMIC subgroup (every value the reference's API can produce is there: Gt::one, pairing(), products, powers and
// inverses of such, lib.rs:165-183): f^-1 = conj(f) is free and a squaring is Granger-Scott's (6 instead of 12 Fq2 products), so
// the chain is 4-bit SIGNED (Booth) windows over the nine entries a^0 .. a^8 - 252 cyclotomic squarings + 64 products + a table of
// 4 squarings and 3 products: 37 % fewer Fq2 products than the general chain above and a table of 9 instead of 16 entries.  Same
// field element as fields/mod.rs:35-46, hence the same bytes.  The caller has checked membership (gt_is_cyclotomic).
BN_FN int booth_digit_256(const uint32_t *k, int i) {          // radix-16 Booth digit i of an 8-word scalar below 2^255, in [-8, 8]
    const int pos = 4 * i - 1;
    uint32_t x;
    if (pos < 0) {
        x = (k[0] << 1) & 31u;
    } else {
        const int w = pos >> 5, sh = pos & 31;
        uint64_t two = (uint64_t)k[w] | (w + 1 < 8 ? ((uint64_t)k[w + 1] << 32) : 0);
        x = (uint32_t)(two >> sh) & 31u;
    }
    return (int)((x >> 1) & 7u) + (int)(x & 1u) - (int)((x >> 4) << 3);
}
// conj(f) for the lane pairs with `neg` set (the digits differ per pair)
template <class F2>
BN_FN Fq6<F2> f6_cond_neg(bool neg, const Fq6<F2> &v) {
    const Fq6<F2> n = f6_neg(v);
    return {f2_select(neg, v.c0, n.c0), f2_select(neg, v.c1, n.c1), f2_select(neg, v.c2, n.c2)};
}
template <class F2, class Tbl>
BN_FN Fq12<F2> gt_pow_cyclotomic(const Fq12<F2> &base, const uint32_t *k_raw, Tbl &tbl) {
    tbl.put(0, f12_one<F2>());
    tbl.put(1, base);
#pragma unroll 1
    for (int e = 2; e <= 8; ++e) {                 // a^e = (a^(e/2))^2 for even e, a^(e-1) * a for odd e
        Fq12<F2> v;
        if ((e & 1) == 0) v = f12_cyclotomic_sqr(Fq12<F2>{tbl.c0(e >> 1), tbl.c1(e >> 1)});
        else v = f12_mul_src(Fq12<F2>{tbl.c0(e - 1), tbl.c1(e - 1)}, Fq12Slot<F2, Tbl>{tbl, 1}, false);
        tbl.put(e, v);
    }
    Fq12<F2> res = f12_one<F2>();
#pragma unroll 1
    for (int w = 63; w >= 0; --w) {
        BN_EXP_HOOK(63 - w, 64);
        if (w != 63) {
#pragma unroll 1
            for (int d = 0; d < 4; ++d) res = f12_cyclotomic_sqr(res);
        }
        const int digit = booth_digit_256(k_raw, w);                            // per lane pair: both lanes hold the same scalar
        // res * conj(t) = conj(conj(res) * t): the sign is applied to the running value (in registers anyway), so the table operand
        // takes the same path as in the exponentiation machine (conjugating the operand on load cost 97 spilled VGPRs)
        res.c1 = f6_cond_neg(digit < 0, res.c1);
        res = f12_mul_src(res, Fq12Slot<F2, Tbl>{tbl, digit < 0 ? -digit : digit}, false);
        res.c1 = f6_cond_neg(digit < 0, res.c1);
    }
    return res;
}
// a in the cyclotomic subgroup of Fq12, i.e. a^(q^4 - q^2 + 1) = 1  <=>  frob^4(a) * a == frob^2(a)   (two Frobenius maps, one product)
template <class F2>
BN_FN bool gt_is_cyclotomic(const Fq12<F2> &a) {
    const Fq12<F2> a2 = f12_frobenius_one<2>(a);
    const Fq12<F2> a4 = f12_frobenius_one<2>(a2);
    const Fq12<F2> l = f12_mul_o(a4, a);
    return f2_is_zero(f2_sub<1, 4>(l.c0.c0, a2.c0.c0)) & f2_is_zero(f2_sub<1, 4>(l.c0.c1, a2.c0.c1)) & f2_is_zero(f2_sub<1, 4>(l.c0.c2, a2.c0.c2)) &
           f2_is_zero(f2_sub<1, 4>(l.c1.c0, a2.c1.c0)) & f2_is_zero(f2_sub<1, 4>(l.c1.c1, a2.c1.c1)) & f2_is_zero(f2_sub<1, 4>(l.c1.c2, a2.c1.c2));
}
template <class F2>
struct PowTableVars {
    Fq12<F2> s_[33];
    BN_FN void put(int i, const Fq12<F2> &v) { s_[i] = v; }
    BN_FN Fq6<F2> c0(int i) const { return s_[i].c0; }
    BN_FN Fq6<F2> c1(int i) const { return s_[i].c1; }
};

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


// Prologue of a pairing in the lane-pair mapping: both Jacobian -> affine conversions with ONE Fermat chain per pair
// (the even lane inverts z_P, the odd lane inverts norm(z_Q) = z0^2 + z1^2; the results are swapped by DPP).
template <class T>
BN_FN void pair_prologue(const T &px, const T &py, const T &pz, const Fq2B<T> &qx, const Fq2B<T> &qy, const Fq2B<T> &qz,
                         G1Aff<T> &p, G2Aff<Fq2B<T>> &q) {
    T sq = fe_sqr(qz.v);
    T n = fe_lc3<1, 1, 0>(sq, lane_partner(sq), sq);                 // norm(z_Q), same in both lanes
    T inv = fe_inverse(lane_pick(pz, n));
    T pinv = lane_partner(inv);
    T izp = lane_pick(inv, pinv), in = lane_pick(pinv, inv);          // 1/z_P and 1/norm in both lanes
    T zi2p = fe_sqr(izp);
    p.x = fe_mul(px, zi2p);
    p.y = fe_mul(py, fe_mul(zi2p, izp));
    T r = fe_mul(qz.v, in);
    Fq2B<T> zi = {lane_pick(r, fe_lc3<-1, 0, 0>(r, r, r))};           // conj(z_Q) / norm
    Fq2B<T> zi2 = f2_sqr(zi);
    q.x = f2_mul(qx, zi2);
    q.y = f2_mul(qy, f2_mul(zi2, zi));
}

#undef F2P
}  // namespace bn254

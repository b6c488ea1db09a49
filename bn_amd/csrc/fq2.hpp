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
BN_FN Fq2A f2_ssub(const Fq2A &a, const Fq2A &b) { return {fe_ssub(a.c0, b.c0), fe_ssub(a.c1, b.c1)}; }
BN_FN Fq2A f2_norm(const Fq2A &a) { return {fe_norm(a.c0), fe_norm(a.c1)}; }
BN_FN Fq2A f2_std(const Fq2A &a) { return {fe_std(a.c0), fe_std(a.c1)}; }
// componentwise reduce(C1*x + C2*y + C3*z)
template <int C1, int C2, int C3>
BN_FN Fq2A f2_lc3(const Fq2A &x, const Fq2A &y, const Fq2A &z) {
    return {fe_lc3<C1, C2, C3>(x.c0, y.c0, z.c0), fe_lc3<C1, C2, C3>(x.c1, y.c1, z.c1)};
}
template <int C1, int C2, int C3>
BN_FN Fq2A f2_lc3w(const Fq2A &x, const Fq2A &y, const Fq2A &z) {
    return {fe_lc3w<C1, C2, C3>(x.c0, y.c0, z.c0), fe_lc3w<C1, C2, C3>(x.c1, y.c1, z.c1)};
}
BN_FN Fq2A f2_sum_for_mul(const Fq2A &a, const Fq2A &b) { return f2_lc3<1, 1, 0>(a, b, b); }     // Karatsuba needs it reduced
BN_FN Fq2A f2_sum3_for_mul(const Fq2A &a, const Fq2A &b, const Fq2A &c) { return f2_lc3<1, 1, 1>(a, b, c); }
BN_FN Fq2A f2_neg(const Fq2A &a) { return f2_lc3<-1, 0, 0>(a, a, a); }
BN_FN Fq2A f2_conj(const Fq2A &a) { return {a.c0, fe_lc3<-1, 0, 0>(a.c1, a.c1, a.c1)}; }
BN_FN Fq2A f2_zero(const Fq2A *) { return {fe_zero(), fe_zero()}; }
BN_FN Fq2A f2_one(const Fq2A *) { return {fe_one(), fe_zero()}; }
template <class T>
BN_FN Fq2A f2_const(const Fq2A *, const T &tab) { return {fe_const(tab[0]), fe_const(tab[1])}; }
BN_FN bool f2_is_zero(const Fq2A &a) { return fe_is_zero(a.c0) & fe_is_zero(a.c1); }
BN_FN bool f2_is_zero_std(const Fq2A &a) { return fe_is_zero_std(a.c0) & fe_is_zero_std(a.c1); }
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
// a / 2: lazy result for the first operand of a product; and in the form f2_sqr accepts (this mapping's squaring wants vb <= 3,
// so it keeps the product by 2^-1)
BN_FN Fq2A f2_half(const Fq2A &a) { return {fe_half(a.c0), fe_half(a.c1)}; }
BN_FN Fq2A f2_half_for_sqr(const Fq2A &a) { return f2_scale(a, fe_const(k::TWO_INV)); }
// reduce(CX * xi * x + CY * y), xi = 9 + i:  xi*x = (9 x0 - x1) + (9 x1 + x0) i.   x, y may be lazy.
template <int CX, int CY>
BN_FN Fq2A f2_lc_xi(const Fq2A &x, const Fq2A &y) {
    return {fe_lc3<9 * CX, -CX, CY>(x.c0, x.c1, y.c0), fe_lc3<9 * CX, CX, CY>(x.c1, x.c0, y.c1)};
}
// reduce(CX*xi*x + CY*y + CZ*z)   (inline: four operands do not fit the register-passing convention of the leaves)
template <int CX, int CY, int CZ>
BN_FN Fq2A f2_lc_xi2(const Fq2A &x, const Fq2A &y, const Fq2A &z) {
    return {fe_lc4_core<9 * CX, -CX, CY, CZ>(x.c0, x.c1, y.c0, z.c0, false), fe_lc4_core<9 * CX, CX, CY, CZ>(x.c1, x.c0, y.c1, z.c1, false)};
}
// every term in the 64-bit chain (fe.hpp WIDE): signed lazy inputs of any limb bound <= 4 each
template <int CX, int CY, int CZ>
BN_FN Fq2A f2_lc_xi2w(const Fq2A &x, const Fq2A &y, const Fq2A &z) {
    return {fe_lc4_core<9 * CX, -CX, CY, CZ, true>(x.c0, x.c1, y.c0, z.c0, false), fe_lc4_core<9 * CX, CX, CY, CZ, true>(x.c1, x.c0, y.c1, z.c1, false)};
}
template <int C1, int C2, int C3>
BN_FN Fq2A f2_lc3sw(const Fq2A &x, const Fq2A &y, const Fq2A &z) {
    return {fe_lc4_core<C1, C2, C3, 0, true>(x.c0, y.c0, z.c0, z.c0, false), fe_lc4_core<C1, C2, C3, 0, true>(x.c1, y.c1, z.c1, z.c1, false)};
}
BN_FN Fq2A f2_mul_xi(const Fq2A &x) { return f2_lc_xi<1, 0>(x, x); }
// (27 - 3i) * x = (27 x0 + 3 x1) + (27 x1 - 3 x0) i: the curve constant 3 b' t^6 of the isomorphic curve (pairing.hpp)
BN_FN Fq2A f2_mul_iso3b(const Fq2A &x) { return {fe_lc3<27, 3, 0>(x.c0, x.c1, x.c1), fe_lc3<27, -3, 0>(x.c1, x.c0, x.c0)}; }
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
template <class TAB> BN_FN Fe f2_scalar_const(const Fq2A *, const TAB &tab) { return fe_const(tab); }


// Karatsuba: a_i b_j + a_j b_i as a signed lazy sum (limb bound 3), given p_ii = a_i b_i and p_jj = a_j b_j (fq6.rs:26-39 forms it the same way)
BN_FN Fq2A f2_cross(const Fq2A &ai, const Fq2A &aj, const Fq2A &bi, const Fq2A &bj, const Fq2A &pii, const Fq2A &pjj) {
    return f2_ssub(f2_ssub(f2_mul(f2_add(ai, aj), f2_norm(f2_add(bi, bj))), pii), pjj);
}

// ===================================================================================================== policy B
// One Fq2 element per PAIR of adjacent lanes (even lane: c0, odd lane: c1).  T is the per-lane scalar: Fe on the GPU;
// the host simulation instantiates it with a 2-lane value type so the same code is checked on the CPU.
// Lane primitives (defined per build): lane_odd<T>(), lane_partner(x), lane_pick(even_choice, odd_choice), lane_bcast<T>(Fe).
#if !defined(BN_HOSTSIM)
BN_FN bool lane_is_odd() { return (threadIdx.x & 1u) != 0; }
BN_FN Fe lane_partner(const Fe &x) {          // DPP quad_perm [1,0,3,2]: swap the two lanes of every pair, no LDS, no memory
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[i], 0xB1, 0xF, 0xF, true);
    return r;
}
BN_FN bool lane_partner_flag(bool f) { return __builtin_amdgcn_mov_dpp((int)f, 0xB1, 0xF, 0xF, true) != 0; }
BN_FN Fe lane_pick(const Fe &even_choice, const Fe &odd_choice) { return fe_select(lane_is_odd(), even_choice, odd_choice); }
BN_FN Fe lane_bcast(const Fe *, const Fe &x) { return x; }
// An Fq2 constant (c0 on even lanes, c1 on odd lanes) built from literals at the point of use: c0 ^ (mask & (c0 ^ c1)), two
// full-rate instructions per limb.  The mask goes through an empty volatile asm so that the value is NOT loop-invariant for
// the compiler: hoisted out of the Miller loop it was kept as nine partially built vectors in private memory (81 dword loads
// per doubling step).
template <class TAB>
BN_FN Fe lane_const_pick(const Fe *, const TAB &even_tab, const TAB &odd_tab) {
    uint32_t m = 0u - (threadIdx.x & 1u);
    asm volatile("" : "+v"(m));
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = even_tab[i] ^ (m & (even_tab[i] ^ odd_tab[i]));
    BN_SETB(r, 1, 1);
    return r;
}
BN_FN Fe lane_load_pair(const Fe *, const uint32_t *w0, const uint32_t *w1) { return fe_from_u32x8(lane_is_odd() ? w1 : w0); }
BN_FN void lane_store_pair(const Fe &a, uint32_t *w0, uint32_t *w1) { fe_to_u32x8(a, lane_is_odd() ? w1 : w0); }
BN_FN bool lane_pair_all_zero(const Fe &a) { bool z = fe_is_zero(a); return z && lane_partner_flag(z); }
BN_FN bool lane_pair_all_zero_std(const Fe &a) { bool z = fe_is_zero_std(a); return z && lane_partner_flag(z); }
// reduce(C1*x + s*C2*y + C3*z), s = -1 on even lanes, +1 on odd lanes
template <int C1, int C2, int C3>
BN_FN Fe fe_lc3_par_body(const Fe &x, const Fe &y, const Fe &z) { return fe_lc3_core<C1, C2, C3, true>(x, y, z, !lane_is_odd()); }
BN_LEAF3T(fe_lc3_par, fe_lc3_par_body)
template <int C1, int C2, int C3, int C4>
BN_FN Fe fe_lc4_par(const Fe &x, const Fe &y, const Fe &z, const Fe &w) { return fe_lc4_core<C1, C2, C3, C4, false, true>(x, y, z, w, !lane_is_odd()); }
template <int C1, int C2, int C3, int C4>      // all terms in the 64-bit chain (fe.hpp WIDE)
BN_FN Fe fe_lc4w_par(const Fe &x, const Fe &y, const Fe &z, const Fe &w) { return fe_lc4_core<C1, C2, C3, C4, true>(x, y, z, w, !lane_is_odd()); }
template <int C1, int C2, int C3>
BN_FN Fe fe_lc3sw(const Fe &x, const Fe &y, const Fe &z) { return fe_lc4_core<C1, C2, C3, 0, true>(x, y, z, z, false); }
#endif

template <class T>
struct Fq2B {
    T v;                 // this lane's component
    using Scalar = T;    // an Fq replicated in both lanes of the pair
};
#define TP ((const T *)nullptr)

// the product leaf: own_a*u + partner_a*v with one reduction (see the header comment).  a: lb <= 2, vb <= 6;  b: S
template <class T>
BN_FN T f2b_mul_body(const T &a, const T &b) {
    T pa = lane_partner(a), pb = lane_partner(b);
    T u = lane_pick(b, pb);
    T v = lane_pick(fe_neg<1, 9>(pb), b);
    return fe_mul2(a, u, pa, v);
}
// complex squaring: even lane (a0+a1)(a0-a1), odd lane (2 a0) a1 : ONE plain Montgomery product per lane
template <class T>
BN_FN T f2b_sqr_body(const T &a) {
    T pa = lane_partner(a);
    T s = lane_pick(fe_add(a, pa), fe_dbl(pa));
    T t = lane_pick(fe_sub<1, 7>(a, pa), a);
    return fe_mul_body(s, t);
}
// the product of two SIGNED differences (fe_sdiff values: |limb| < 2^29) - a Karatsuba cross product (a_i - a_j)(b_j - b_i) of tower.hpp,
// which needs no carry propagation of its operands; the result is a signed lazy value for the fused reductions
template <class T>
BN_FN T f2b_muls_body(const T &a, const T &b) {
    T pa = lane_partner(a), pb = lane_partner(b);
    T u = lane_pick(b, pb);
    T v = lane_pick(fe_sneg(pb), b);
    return fe_mul2s(a, u, pa, v);
}
#if !defined(BN_HOSTSIM)
// The GPU's operand set-up for the two leaves above, nine instructions per role decision instead of eighteen: where only the EVEN lane
// of a pair differs from the odd one, the difference is applied IN PLACE to a register the exchange just produced, with the odd lanes
// switched off in EXEC for those instructions (one asm statement: save, mask, nine limbs, restore) - not computed in all lanes and then
// selected.  Same values limb for limb as f2b_mul_body / f2b_sqr_body (which the host simulation keeps running):
//   product: this lane's component = a0 * (own b) + a1 * X,   a0 / a1 = the pair's components of a in BOTH lanes (quad_perm [0,0,2,2] /
//            [1,1,3,3]),  X = the partner's b, negated on the even lane          (45 -> 36 instructions around the 243 multiply-adds)
//   square:  S = own a doubled, T = the partner's a;  even lane: S = a + T, T = a - T  (63 -> 45 around the 162 multiply-adds)
// (the same exchanges through the LDS crossbar - ds_swizzle in quad-permute mode - measured 5 % slower: profiles/r04r_ab_swizzle_exchange.txt)
BN_FN Fe lane_dpp_even(const Fe &x) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[i], 0xA0, 0xF, 0xF, true);
    return r;
}
BN_FN Fe lane_dpp_odd(const Fe &x) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[i], 0xF5, 0xF, 0xF, true);
    return r;
}
BN_FN Fe lane_partner_x(const Fe &x) { return lane_partner(x); }
#define BN_EVEN_LANES 0x5555555555555555ull
template <int LB, int K>
BN_FN void fe_neg_on_even_lanes(Fe &x) {            // x <- fe_neg<LB, K>(x) on the even lanes, untouched on the odd ones
    constexpr Bias<LB, K> B{};
    uint64_t saved;
    asm("s_and_saveexec_b64 %[sv], %[m]\n\t"
        "v_sub_u32 %[x0], %[b0], %[x0]\n\tv_sub_u32 %[x1], %[b1], %[x1]\n\tv_sub_u32 %[x2], %[b2], %[x2]\n\t"
        "v_sub_u32 %[x3], %[b3], %[x3]\n\tv_sub_u32 %[x4], %[b4], %[x4]\n\tv_sub_u32 %[x5], %[b5], %[x5]\n\t"
        "v_sub_u32 %[x6], %[b6], %[x6]\n\tv_sub_u32 %[x7], %[b7], %[x7]\n\tv_sub_u32 %[x8], %[b8], %[x8]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [x0] "+v"(x.l[0]), [x1] "+v"(x.l[1]), [x2] "+v"(x.l[2]), [x3] "+v"(x.l[3]), [x4] "+v"(x.l[4]), [x5] "+v"(x.l[5]), [x6] "+v"(x.l[6]),
          [x7] "+v"(x.l[7]), [x8] "+v"(x.l[8]), [sv] "=&s"(saved)
        : [b0] "s"(B.c[0]), [b1] "s"(B.c[1]), [b2] "s"(B.c[2]), [b3] "s"(B.c[3]), [b4] "s"(B.c[4]), [b5] "s"(B.c[5]), [b6] "s"(B.c[6]),
          [b7] "s"(B.c[7]), [b8] "s"(B.c[8]), [m] "s"(BN_EVEN_LANES)
        : "scc");
}
BN_FN Fe f2b_mul_gpu(const Fe &a, const Fe &b) {
    const Fe a0 = lane_dpp_even(a), a1 = lane_dpp_odd(a);
    Fe x = lane_partner_x(b);
    fe_neg_on_even_lanes<1, 9>(x);
    return fe_mul2(a0, b, a1, x);
}
BN_FN Fe f2b_sqr_gpu(const Fe &a) {
    constexpr Bias<1, 7> B{};
    Fe t = lane_partner_x(a), s = fe_dbl(a);
    uint64_t saved;
#define BN_SQ(i) "v_add_u32 %[s" #i "], %[a" #i "], %[t" #i "]\n\tv_sub_u32 %[t" #i "], %[a" #i "], %[t" #i "]\n\tv_add_u32 %[t" #i "], %[b" #i "], %[t" #i "]\n\t"
    asm("s_and_saveexec_b64 %[sv], %[m]\n\t" BN_SQ(0) BN_SQ(1) BN_SQ(2) BN_SQ(3) BN_SQ(4) BN_SQ(5) BN_SQ(6) BN_SQ(7) BN_SQ(8) "s_mov_b64 exec, %[sv]"
        : [s0] "+v"(s.l[0]), [s1] "+v"(s.l[1]), [s2] "+v"(s.l[2]), [s3] "+v"(s.l[3]), [s4] "+v"(s.l[4]), [s5] "+v"(s.l[5]), [s6] "+v"(s.l[6]),
          [s7] "+v"(s.l[7]), [s8] "+v"(s.l[8]),
          [t0] "+v"(t.l[0]), [t1] "+v"(t.l[1]), [t2] "+v"(t.l[2]), [t3] "+v"(t.l[3]), [t4] "+v"(t.l[4]), [t5] "+v"(t.l[5]), [t6] "+v"(t.l[6]),
          [t7] "+v"(t.l[7]), [t8] "+v"(t.l[8]), [sv] "=&s"(saved)
        : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]), [a6] "v"(a.l[6]),
          [a7] "v"(a.l[7]), [a8] "v"(a.l[8]),
          [b0] "s"(B.c[0]), [b1] "s"(B.c[1]), [b2] "s"(B.c[2]), [b3] "s"(B.c[3]), [b4] "s"(B.c[4]), [b5] "s"(B.c[5]), [b6] "s"(B.c[6]),
          [b7] "s"(B.c[7]), [b8] "s"(B.c[8]), [m] "s"(BN_EVEN_LANES)
        : "scc");
#undef BN_SQ
    return fe_mul_body(s, t);
}
BN_FN Fe f2b_muls_gpu(const Fe &a, const Fe &b) {        // signed operands: the even lane's negation is a plain 0 - x
    const Fe a0 = lane_dpp_even(a), a1 = lane_dpp_odd(a);
    Fe x = lane_partner_x(b);
    uint64_t saved;
    asm("s_and_saveexec_b64 %[sv], %[m]\n\t"
        "v_sub_u32 %[x0], 0, %[x0]\n\tv_sub_u32 %[x1], 0, %[x1]\n\tv_sub_u32 %[x2], 0, %[x2]\n\t"
        "v_sub_u32 %[x3], 0, %[x3]\n\tv_sub_u32 %[x4], 0, %[x4]\n\tv_sub_u32 %[x5], 0, %[x5]\n\t"
        "v_sub_u32 %[x6], 0, %[x6]\n\tv_sub_u32 %[x7], 0, %[x7]\n\tv_sub_u32 %[x8], 0, %[x8]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [x0] "+v"(x.l[0]), [x1] "+v"(x.l[1]), [x2] "+v"(x.l[2]), [x3] "+v"(x.l[3]), [x4] "+v"(x.l[4]), [x5] "+v"(x.l[5]), [x6] "+v"(x.l[6]),
          [x7] "+v"(x.l[7]), [x8] "+v"(x.l[8]), [sv] "=&s"(saved)
        : [m] "s"(BN_EVEN_LANES)
        : "scc");
    return fe_mul2s(a0, b, a1, x);
}
BN_LEAF_MUL u32x9 f2b_muls_leaf(u32x9 a, u32x9 b) { return bn_tov(f2b_muls_gpu(bn_unv(a), bn_unv(b))); }
BN_FN Fe f2b_muls(const Fe &a, const Fe &b) { return bn_unv(f2b_muls_leaf(bn_tov(a), bn_tov(b))); }
BN_FN Fe f2b_mul_inl(const Fe &a, const Fe &b) { return f2b_mul_gpu(a, b); }      // for callers that inline the leaf themselves (wave.hpp)
BN_FN Fe f2b_sqr_inl(const Fe &a) { return f2b_sqr_gpu(a); }
BN_LEAF_MUL u32x9 f2b_mul_leaf(u32x9 a, u32x9 b) { return bn_tov(f2b_mul_gpu(bn_unv(a), bn_unv(b))); }
BN_LEAF_MUL u32x9 f2b_sqr_leaf(u32x9 a) { return bn_tov(f2b_sqr_gpu(bn_unv(a))); }
BN_FN Fe f2b_mul(const Fe &a, const Fe &b) { return bn_unv(f2b_mul_leaf(bn_tov(a), bn_tov(b))); }
BN_FN Fe f2b_sqr(const Fe &a) { return bn_unv(f2b_sqr_leaf(bn_tov(a))); }
#else
template <class T> BN_FN T f2b_mul(const T &a, const T &b) { return f2b_mul_body(a, b); }
template <class T> BN_FN T f2b_sqr(const T &a) { return f2b_sqr_body(a); }
template <class T> BN_FN T f2b_mul_inl(const T &a, const T &b) { return f2b_mul_body(a, b); }
template <class T> BN_FN T f2b_sqr_inl(const T &a) { return f2b_sqr_body(a); }
template <class T> BN_FN T f2b_muls(const T &a, const T &b) { return f2b_muls_body(a, b); }
#endif

// A multiplier PREPARED for the lane-pair product: (u, v) such that this lane's component of a * b is own_a * u + partner_a * v.  With
// every limb normalized (the negated component too), three products can share one reduction (fe_mul6):
//     reduce(a1 b1 + a2 b2 + a3 b3)     inputs: normalized limbs (standard form, or carry-propagated sums below 4q for the a_i)
// which is how the sparse line product forms each of its six output coefficients (tower.hpp f12_mul_by_024).
template <class T> struct Fq2BPrep { T u, v; };
template <class T>
BN_FN Fq2BPrep<T> f2b_prepare(const Fq2B<T> &b) {
    const T pb = lane_partner(b.v);
    return {lane_pick(b.v, pb), lane_pick(fe_norm(fe_neg<1, 9>(pb)), b.v)};
}
#if !defined(BN_HOSTSIM)
// the GPU's set-up (same limbs): b0 and b1 by quad_perm broadcasts, the even lane's negation in place under the EXEC mask, ONE carry
// propagation over all lanes (the odd lane's b1 is normalized already and passes through unchanged): 54 instead of 63 instructions
BN_FN Fq2BPrep<Fe> f2b_prepare(const Fq2B<Fe> &b) {
    const Fe u = lane_dpp_even(b.v);
    Fe t = lane_dpp_odd(b.v);
    fe_neg_on_even_lanes<1, 9>(t);
    return {u, fe_norm(t)};
}
#endif
template <class T>
BN_FN Fq2B<T> f2b_mul3(const Fq2B<T> &a1, const Fq2BPrep<T> &b1, const Fq2B<T> &a2, const Fq2BPrep<T> &b2, const Fq2B<T> &a3, const Fq2BPrep<T> &b3) {
    return {fe_mul6(a1.v, b1.u, lane_partner(a1.v), b1.v, a2.v, b2.u, lane_partner(a2.v), b2.v, a3.v, b3.u, lane_partner(a3.v), b3.v)};
}
// reduce(a1 b1 + a2 b2 + a3 s), s in Fq replicated in both lanes of the pair (fe_mul5: the third product is this lane's component times s)
template <class T>
BN_FN Fq2B<T> f2b_mul3s(const Fq2B<T> &a1, const Fq2BPrep<T> &b1, const Fq2B<T> &a2, const Fq2BPrep<T> &b2, const Fq2B<T> &a3, const T &s) {
    return {fe_mul5(a1.v, b1.u, lane_partner(a1.v), b1.v, a2.v, b2.u, lane_partner(a2.v), b2.v, a3.v, s)};
}
template <class T> BN_FN Fq2B<T> f2_add(const Fq2B<T> &a, const Fq2B<T> &b) { return {fe_add(a.v, b.v)}; }
template <class T> BN_FN Fq2B<T> f2_dbl(const Fq2B<T> &a) { return {fe_dbl(a.v)}; }
template <int LB, int K, class T> BN_FN Fq2B<T> f2_sub(const Fq2B<T> &a, const Fq2B<T> &b) { return {fe_sub<LB, K>(a.v, b.v)}; }
template <class T> BN_FN Fq2B<T> f2_ssub(const Fq2B<T> &a, const Fq2B<T> &b) { return {fe_ssub(a.v, b.v)}; }
template <class T> BN_FN Fq2B<T> f2_sdiff(const Fq2B<T> &a, const Fq2B<T> &b) { return {fe_sdiff(a.v, b.v)}; }      // operands of f2_muls
template <class T> BN_FN Fq2B<T> f2_muls(const Fq2B<T> &a, const Fq2B<T> &b) { return {f2b_muls(a.v, b.v)}; }
template <class T> BN_FN Fq2B<T> f2_norm(const Fq2B<T> &a) { return {fe_norm(a.v)}; }
template <class T> BN_FN Fq2B<T> f2_std(const Fq2B<T> &a) { return {fe_std(a.v)}; }
template <int C1, int C2, int C3, class T>
BN_FN Fq2B<T> f2_lc3(const Fq2B<T> &x, const Fq2B<T> &y, const Fq2B<T> &z) { return {fe_lc3<C1, C2, C3>(x.v, y.v, z.v)}; }
template <int C1, int C2, int C3, class T>
BN_FN Fq2B<T> f2_lc3w(const Fq2B<T> &x, const Fq2B<T> &y, const Fq2B<T> &z) { return {fe_lc3w<C1, C2, C3>(x.v, y.v, z.v)}; }
template <class T> BN_FN Fq2B<T> f2_sum_for_mul(const Fq2B<T> &a, const Fq2B<T> &b) { return {fe_norm(fe_add(a.v, b.v))}; }
template <class T> BN_FN Fq2B<T> f2_sum3_for_mul(const Fq2B<T> &a, const Fq2B<T> &b, const Fq2B<T> &c) { return {fe_norm(fe_add(fe_add(a.v, b.v), c.v))}; }
template <class T> BN_FN Fq2B<T> f2_neg(const Fq2B<T> &a) { return {fe_lc3<-1, 0, 0>(a.v, a.v, a.v)}; }
template <class T> BN_FN Fq2B<T> f2_conj(const Fq2B<T> &a) { return {lane_pick(a.v, fe_lc3<-1, 0, 0>(a.v, a.v, a.v))}; }
template <class T> BN_FN Fq2B<T> f2_neg_lazy(const Fq2B<T> &a) { return {fe_neg<1, 4>(a.v)}; }
template <class T> BN_FN Fq2B<T> f2_conj_lazy(const Fq2B<T> &a) { return {lane_pick(a.v, fe_neg<1, 4>(a.v))}; }
template <class T> BN_FN Fq2B<T> f2_zero(const Fq2B<T> *) { return {lane_bcast(TP, fe_zero())}; }
template <class T> BN_FN Fq2B<T> f2_one(const Fq2B<T> *) { return {lane_pick(lane_bcast(TP, fe_one()), lane_bcast(TP, fe_zero()))}; }
template <class T, class TAB>
BN_FN Fq2B<T> f2_const(const Fq2B<T> *, const TAB &tab) { return {lane_const_pick(TP, tab[0], tab[1])}; }
template <class T> BN_FN Fq2B<T> f2_select(bool take_b, const Fq2B<T> &a, const Fq2B<T> &b) { return {fe_select(take_b, a.v, b.v)}; }
template <class T> BN_FN Fq2B<T> f2_mul(const Fq2B<T> &a, const Fq2B<T> &b) { return {f2b_mul(a.v, b.v)}; }
// Karatsuba: a_i b_j + a_j b_i = (a_i - a_j)(b_j - b_i) + a_i b_i + a_j b_j, as a signed lazy sum (limb bound 3).  The differences of
// normalized operands are signed values of limb magnitude below 2^29 and go into the signed dual product as they are, where the sums
// (a_i + a_j)(b_i + b_j) cost a carry propagation each (27 instructions; profiles/r04n_ab_signed_karatsuba.txt).
template <class T>
BN_FN Fq2B<T> f2_cross(const Fq2B<T> &ai, const Fq2B<T> &aj, const Fq2B<T> &bi, const Fq2B<T> &bj, const Fq2B<T> &pii, const Fq2B<T> &pjj) {
    return f2_add(f2_add(f2_muls(f2_sdiff(ai, aj), f2_sdiff(bj, bi)), pii), pjj);
}
template <class T> BN_FN Fq2B<T> f2_sqr(const Fq2B<T> &a) { return {f2b_sqr(a.v)}; }
template <class T> BN_FN Fq2B<T> f2_scale(const Fq2B<T> &a, const T &s) { return {fe_mul(a.v, s)}; }
template <class T> BN_FN Fq2B<T> f2_half(const Fq2B<T> &a) { return {fe_half(a.v)}; }
template <class T> BN_FN Fq2B<T> f2_half_for_sqr(const Fq2B<T> &a) { return {fe_norm(fe_half(a.v))}; }      // normalized limbs, vb <= 6
// reduce(CX*xi*x + CY*y): even lane 9CX*x0 - CX*x1 + CY*y0, odd lane 9CX*x1 + CX*x0 + CY*y1
template <int CX, int CY, class T>
BN_FN Fq2B<T> f2_lc_xi(const Fq2B<T> &x, const Fq2B<T> &y) { return {fe_lc3_par<9 * CX, CX, CY>(x.v, lane_partner(x.v), y.v)}; }
template <int CX, int CY, int CZ, class T>
BN_FN Fq2B<T> f2_lc_xi2(const Fq2B<T> &x, const Fq2B<T> &y, const Fq2B<T> &z) { return {fe_lc4_par<9 * CX, CX, CY, CZ>(x.v, lane_partner(x.v), y.v, z.v)}; }
// the same with every term in the 64-bit chain, and the plain three-term combination likewise: signed lazy inputs of any limb bound <= 4 each
template <int CX, int CY, int CZ, class T>
BN_FN Fq2B<T> f2_lc_xi2w(const Fq2B<T> &x, const Fq2B<T> &y, const Fq2B<T> &z) { return {fe_lc4w_par<9 * CX, CX, CY, CZ>(x.v, lane_partner(x.v), y.v, z.v)}; }
template <int C1, int C2, int C3, class T>
BN_FN Fq2B<T> f2_lc3sw(const Fq2B<T> &x, const Fq2B<T> &y, const Fq2B<T> &z) { return {fe_lc3sw<C1, C2, C3>(x.v, y.v, z.v)}; }
template <class T> BN_FN Fq2B<T> f2_mul_xi(const Fq2B<T> &x) { return f2_lc_xi<1, 0>(x, x); }
// (27 - 3i) * x: even lane 27 x0 + 3 x1, odd lane 27 x1 - 3 x0  (fe_lc3_par negates its middle term on the even lane)
template <class T> BN_FN Fq2B<T> f2_mul_iso3b(const Fq2B<T> &x) { return {fe_lc3_par<27, -3, 0>(x.v, lane_partner(x.v), x.v)}; }
template <class T, class TAB>
BN_FN Fq2B<T> f2_mul_const(const Fq2B<T> &a, const TAB &tab) { return f2_mul(a, f2_const((const Fq2B<T> *)nullptr, tab)); }
// fq2.rs:125-136: norm = a0^2 + a1^2 (one square per lane, exchanged), ONE Fermat chain for the pair
template <class T>
BN_FN Fq2B<T> f2_inverse(const Fq2B<T> &a) {
    T sq = fe_sqr(a.v);
    T n = fe_lc3<1, 1, 0>(sq, lane_partner(sq), sq);
    T t = fe_inverse(n);
    T r = fe_mul(a.v, t);
    return {lane_pick(r, fe_lc3<-1, 0, 0>(r, r, r))};
}
template <class T> BN_FN bool f2_is_zero(const Fq2B<T> &a) { return lane_pair_all_zero(a.v); }
template <class T> BN_FN bool f2_is_zero_std(const Fq2B<T> &a) { return lane_pair_all_zero_std(a.v); }
template <class T> BN_FN Fq2B<T> f2_load(const Fq2B<T> *, const uint32_t *w) { return {lane_load_pair(TP, w, w + 8)}; }
template <class T> BN_FN void f2_store(const Fq2B<T> &a, uint32_t *w) { lane_store_pair(a.v, w, w + 8); }
template <class T> BN_FN T f2_scalar_load(const Fq2B<T> *, const uint32_t *w) { return lane_load_pair(TP, w, w); }
template <class T, class TAB> BN_FN T f2_scalar_const(const Fq2B<T> *, const TAB &tab) { return lane_bcast(TP, fe_const(tab)); }
#undef TP
}  // namespace bn254

// TEST INFRASTRUCTURE - host simulation of the DEVICE headers (bn_amd/csrc/*.hpp) compiled with g++ for the CPU.
// Purpose: (1) bit-exact comparison of the engine's arithmetic with the oracle where there is no GPU, and
// (2) -DBN_BOUNDS: run-time enforcement of every limb/value bound of the lazy 9x29-bit number system (the engine's control
// flow is data independent, so one simulated pairing exercises every bound check).
// This library is never loaded by the product (bn_amd/); it is not a CPU fallback.
#define BN_HOSTSIM 1
#include "lanepair.hpp"
#include "lanequad.hpp"
#include "../../bn_amd/csrc/io.hpp"
#include "../../bn_amd/csrc/curve.hpp"
#include "../../bn_amd/csrc/io_wire.hpp"
#include "../../bn_amd/csrc/quad.hpp"
#include <cstring>

using namespace bn254;
#define EXPORT extern "C" __attribute__((visibility("default")))
typedef Fq2A F2;

EXPORT int hs_bounds_enabled() {
#ifdef BN_BOUNDS
    return 1;
#else
    return 0;
#endif
}
// out = a*b, a+b, a-b in the reference image (canonical Montgomery radix 2^256), through the engine's lazy arithmetic
EXPORT void hs_fe_mul(const uint32_t *a, const uint32_t *b, uint32_t *o) { fe_to_u32x8(fe_mul(fe_from_u32x8(a), fe_from_u32x8(b)), o); }
EXPORT void hs_fe_add(const uint32_t *a, const uint32_t *b, uint32_t *o) { fe_to_u32x8(fe_add(fe_from_u32x8(a), fe_from_u32x8(b)), o); }
EXPORT void hs_fe_sub(const uint32_t *a, const uint32_t *b, uint32_t *o) { fe_to_u32x8(fe_sub<1, 3>(fe_from_u32x8(a), fe_from_u32x8(b)), o); }
EXPORT void hs_fe_roundtrip(const uint32_t *a, uint32_t *o) { fe_to_u32x8(fe_from_u32x8(a), o); }
EXPORT void hs_fe_inverse(const uint32_t *a, uint32_t *o) { fe_to_u32x8(fe_inverse(fe_from_u32x8(a)), o); }
EXPORT void hs_fe_inverse_fermat(const uint32_t *a, uint32_t *o) { fe_to_u32x8(fe_inverse_fermat(fe_from_u32x8(a)), o); }
EXPORT int hs_fe_is_zero(const uint32_t *a) { return fe_is_zero(fe_from_u32x8(a)); }
// a stress of the lazy forms: ((a+b)+(a+b)) - b - b + 9a ... reduced through lc3, compared with the oracle's canonical ops
EXPORT void hs_fe_lazy_mix(const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *o) {
    Fe x = fe_from_u32x8(a), y = fe_from_u32x8(b), z = fe_from_u32x8(c);
    Fe s = fe_add(fe_add(x, y), fe_add(x, y));                    // 2x + 2y  (4,8)
    Fe d = fe_sub<1, 3>(fe_sub<1, 3>(s, y), z);                   // 2x + y - z (8, 14)
    Fe r = fe_lc3w<9, -1, 3>(d, x, z);                            // 18x + 9y - 9z - x + 3z = 17x + 9y - 6z
    Fe n = fe_norm(fe_add(r, fe_reduce<9>(y)));                   // + 9y
    fe_to_u32x8(fe_mul(n, fe_one()), o);                          // = 17x + 18y - 6z
}
// signed lazy differences through the narrow/wide reduction: 9(x-y-z) - (y-x) + z = 10x - 10y - 8z
EXPORT void hs_fe_signed_mix(const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *o) {
    Fe x = fe_from_u32x8(a), y = fe_from_u32x8(b), z = fe_from_u32x8(c);
    Fe u = fe_ssub(fe_ssub(x, y), z);
    Fe r = fe_lc3<9, -1, 1>(u, fe_ssub(y, x), z);
    Fe r2 = fe_lc3_core<9, 1, 1>(u, fe_ssub(y, x), z, true);     // run-time negated middle term: same value
    fe_to_u32x8(fe_mul(fe_lc3<1, 1, -1>(r, r2, r), fe_one()), o);
}
EXPORT void hs_fe_mul2(const uint32_t *a, const uint32_t *u, const uint32_t *c, const uint32_t *v, uint32_t *o) {
    fe_to_u32x8(fe_mul2(fe_from_u32x8(a), fe_from_u32x8(u), fe_from_u32x8(c), fe_from_u32x8(v)), o);
}

// the signed dual product on differences of standard elements: (a - b)(c - d) + (b - a)(d - a), back to the standard image through a fused reduction
EXPORT void hs_fe_mul2s(const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d, uint32_t *o) {
    Fe x = fe_from_u32x8(a), y = fe_from_u32x8(b), z = fe_from_u32x8(c), w = fe_from_u32x8(d);
    Fe r = fe_mul2s(fe_sdiff(x, y), fe_sdiff(z, w), fe_sdiff(y, x), fe_sdiff(w, x));
    fe_to_u32x8(fe_mul(fe_lc3<1, 0, 0>(r, r, r), fe_one()), o);
}
EXPORT void hs_fq2_mul(const uint32_t *a, const uint32_t *b, uint32_t *o) { f2_store(f2_mul(f2_load((F2 *)0, a), f2_load((F2 *)0, b)), o); }
EXPORT void hs_fq2_sqr(const uint32_t *a, uint32_t *o) { f2_store(f2_sqr(f2_load((F2 *)0, a)), o); }
EXPORT void hs_fq2_mul_xi(const uint32_t *a, uint32_t *o) { f2_store(f2_mul_xi(f2_load((F2 *)0, a)), o); }
EXPORT void hs_fq2_inverse(const uint32_t *a, uint32_t *o) { f2_store(f2_inverse(f2_load((F2 *)0, a)), o); }

EXPORT void hs_fq12_mul(const uint32_t *a, const uint32_t *b, uint32_t *o) { f12_store(f12_mul(f12_load<F2>(a), f12_load<F2>(b)), o); }
EXPORT void hs_fq12_sqr(const uint32_t *a, uint32_t *o) { f12_store(f12_sqr(f12_load<F2>(a)), o); }
EXPORT void hs_fq12_inverse(const uint32_t *a, uint32_t *o) { f12_store(f12_inverse(f12_load<F2>(a)), o); }
EXPORT void hs_fq12_conj(const uint32_t *a, uint32_t *o) { f12_store(f12_conj(f12_load<F2>(a)), o); }
EXPORT void hs_fq12_cyclotomic_sqr(const uint32_t *a, uint32_t *o) { f12_store(f12_cyclotomic_sqr(f12_load<F2>(a)), o); }
EXPORT void hs_fq12_exp_by_neg_z(const uint32_t *a, uint32_t *o) { f12_store(exp_by_neg_z_reference_schedule(f12_load<F2>(a)), o); }
EXPORT void hs_fq12_exp_by_neg_z_naf(const uint32_t *a, uint32_t *o) { f12_store(exp_by_neg_z(f12_load<F2>(a)), o); }
EXPORT void hs_fq12_frobenius(const uint32_t *a, int p, uint32_t *o) {
    Fq12<F2> f = f12_load<F2>(a);
    f12_store(p == 1 ? f12_frobenius<1>(f) : p == 2 ? f12_frobenius<2>(f) : f12_frobenius<3>(f), o);
}
EXPORT void hs_fq12_mul_by_024(const uint32_t *a, const uint32_t *l0, const uint32_t *lvw, const uint32_t *lvv, uint32_t *o) {
    f12_store(f12_mul_by_024(f12_load<F2>(a), f2_load((F2 *)0, l0), f2_load((F2 *)0, lvw), f2_load((F2 *)0, lvv)), o);
}
EXPORT void hs_final_exponentiation(const uint32_t *a, uint32_t *o) { f12_store(final_exponentiation(f12_load<F2>(a)), o); }

// full pairing of Jacobian inputs in the reference layout (groups/mod.rs:764-771): infinity -> one
EXPORT void hs_miller(const uint32_t *g1, const uint32_t *g2, uint32_t *o) {
    G1Aff<Fe> p = g1_to_affine(fe_from_u32x8(g1), fe_from_u32x8(g1 + 8), fe_from_u32x8(g1 + 16));
    G2Aff<F2> q = g2_to_affine(f2_load((F2 *)0, g2), f2_load((F2 *)0, g2 + 16), f2_load((F2 *)0, g2 + 32));
    f12_store(miller_loop(p, q), o);
}
EXPORT void hs_pairing(const uint32_t *g1, const uint32_t *g2, uint32_t *o) {
    bool inf = words_all_zero(g1 + 16, 8) || words_all_zero(g2 + 32, 16);
    G1Aff<Fe> p = g1_to_affine(fe_from_u32x8(g1), fe_from_u32x8(g1 + 8), fe_from_u32x8(g1 + 16));
    G2Aff<F2> q = g2_to_affine(f2_load((F2 *)0, g2), f2_load((F2 *)0, g2 + 16), f2_load((F2 *)0, g2 + 32));
    Fq12<F2> f = final_exponentiation(miller_loop(p, q));
    if (inf) f = f12_one<F2>();
    f12_store(f, o);
}

// G * Fr through the engine: raw Jacobian result of the reference's double-and-add chain, and its normalized image
template <class F, int W>
static void hs_mul_generic(const uint32_t *pt, const uint32_t *k, uint32_t *o, int normalize, typename F::T (*ld)(const uint32_t *), void (*st)(const typename F::T &, uint32_t *)) {
    Jac<F> p = {ld(pt), ld(pt + W), ld(pt + 2 * W)};
    uint32_t raw[8];
    fr_from_mont(k, raw);
    Jac<F> r = normalize == 3 ? scalar_mul_booth_affine<F>(p, raw) : normalize == 2 ? scalar_mul_windowed<F>(p, raw) : scalar_mul_reference_chain<F>(p, raw);
    if (normalize) r = jac_normalize<F>(r);
    st(r.x, o); st(r.y, o + W); st(r.z, o + 2 * W);
}
static Fe ld1(const uint32_t *w) { return fe_from_u32x8(w); }
static void st1(const Fe &a, uint32_t *w) { fe_to_u32x8(a, w); }
static F2 ld2(const uint32_t *w) { return f2_load((F2 *)0, w); }
static void st2(const F2 &a, uint32_t *w) { f2_store(a, w); }
EXPORT void hs_g1_mul(const uint32_t *p, const uint32_t *k, int normalize, uint32_t *o) { hs_mul_generic<FqField, 8>(p, k, o, normalize, ld1, st1); }
// G1 through the GLV chain (what bn254_g1_mul_batch runs), normalized; and the decomposition itself: |k1|, |k2| (5 words each), signs
EXPORT void hs_g1_mul_glv(const uint32_t *pt, const uint32_t *k, uint32_t *o) {
    Jac<FqField> p = {ld1(pt), ld1(pt + 8), ld1(pt + 16)};
    uint32_t raw[8];
    fr_from_mont(k, raw);
    Jac<FqField> r = jac_normalize<FqField>(scalar_mul_glv(p, raw));
    st1(r.x, o); st1(r.y, o + 8); st1(r.z, o + 16);
}
EXPORT void hs_glv_decompose(const uint32_t *k, uint32_t *o) {
    uint32_t raw[8];
    fr_from_mont(k, raw);
    GlvSplit g = glv_decompose(raw);
    for (int i = 0; i < 5; ++i) { o[i] = g.m1[i]; o[6 + i] = g.m2[i]; }
    o[5] = g.neg1; o[11] = g.neg2;
}
EXPORT void hs_g2_mul(const uint32_t *p, const uint32_t *k, int normalize, uint32_t *o) { hs_mul_generic<Fq2Field<F2>, 16>(p, k, o, normalize, ld2, st2); }
EXPORT void hs_fr_from_mont(const uint32_t *k, uint32_t *o) { fr_from_mont(k, o); }

// ---------------------------------------------------------------- lane-pair mapping (Fq2B) executed on a simulated lane pair
typedef Fq2B<FeP> F2B;
EXPORT void hsb_fq2_mul(const uint32_t *a, const uint32_t *b, uint32_t *o) { f2_store(f2_mul(f2_load((F2B *)0, a), f2_load((F2B *)0, b)), o); }
EXPORT void hsb_fq2_sqr(const uint32_t *a, uint32_t *o) { f2_store(f2_sqr(f2_load((F2B *)0, a)), o); }
EXPORT void hsb_fq2_mul_xi(const uint32_t *a, uint32_t *o) { f2_store(f2_mul_xi(f2_load((F2B *)0, a)), o); }
EXPORT void hsb_fq2_inverse(const uint32_t *a, uint32_t *o) { f2_store(f2_inverse(f2_load((F2B *)0, a)), o); }
EXPORT void hsb_fq12_mul(const uint32_t *a, const uint32_t *b, uint32_t *o) { f12_store(f12_mul(f12_load<F2B>(a), f12_load<F2B>(b)), o); }
EXPORT void hsb_fq12_sqr(const uint32_t *a, uint32_t *o) { f12_store(f12_sqr(f12_load<F2B>(a)), o); }
EXPORT void hsb_fq12_inverse(const uint32_t *a, uint32_t *o) { f12_store(f12_inverse(f12_load<F2B>(a)), o); }
EXPORT void hsb_fq12_cyclotomic_sqr(const uint32_t *a, uint32_t *o) { f12_store(f12_cyclotomic_sqr(f12_load<F2B>(a)), o); }
EXPORT void hsb_fq12_frobenius(const uint32_t *a, int p, uint32_t *o) {
    Fq12<F2B> f = f12_load<F2B>(a);
    f12_store(p == 1 ? f12_frobenius<1>(f) : p == 2 ? f12_frobenius<2>(f) : f12_frobenius<3>(f), o);
}
EXPORT void hsb_fq12_mul_by_024(const uint32_t *a, const uint32_t *l0, const uint32_t *lvw, const uint32_t *lvv, uint32_t *o) {
    f12_store(f12_mul_by_024(f12_load<F2B>(a), f2_load((F2B *)0, l0), f2_load((F2B *)0, lvw), f2_load((F2B *)0, lvv)), o);
}
EXPORT void hsb_final_exponentiation(const uint32_t *a, uint32_t *o) { f12_store(final_exponentiation(f12_load<F2B>(a)), o); }
EXPORT void hsb_pairing(const uint32_t *g1, const uint32_t *g2, uint32_t *o) {
    bool inf = words_all_zero(g1 + 16, 8) || words_all_zero(g2 + 32, 16);
    G1Aff<FeP> p; G2Aff<F2B> q;
    pair_prologue<FeP>(f2_scalar_load((F2B *)0, g1), f2_scalar_load((F2B *)0, g1 + 8), f2_scalar_load((F2B *)0, g1 + 16),
                       f2_load((F2B *)0, g2), f2_load((F2B *)0, g2 + 16), f2_load((F2B *)0, g2 + 32), p, q);
    Fq12<F2B> f = final_exponentiation(miller_loop(p, q));
    if (inf) f = f12_one<F2B>();
    f12_store(f, o);
}
// G2 * Fr in the lane-pair mapping (bn254_kernels_mul.hip bn254_g2_mul_M)
static F2B ld2b(const uint32_t *w) { return f2_load((F2B *)0, w); }
static void st2b(const F2B &a, uint32_t *w) { f2_store(a, w); }
EXPORT void hsb_g2_mul(const uint32_t *p, const uint32_t *k, int normalize, uint32_t *o) { hs_mul_generic<Fq2Field<F2B>, 16>(p, k, o, normalize, ld2b, st2b); }
// G + G / G - G with the reference's branches (bn254_kernels_mul.hip add_body), G1 one lane and G2 on a lane pair
template <class F, int W>
static void hs_add_generic(const uint32_t *a, const uint32_t *b, int negate_b, uint32_t *o, typename F::T (*ld)(const uint32_t *), void (*st)(const typename F::T &, uint32_t *)) {
    Jac<F> pa = {ld(a), ld(a + W), ld(a + 2 * W)}, pb = {ld(b), ld(b + W), ld(b + 2 * W)};
    const bool bz = F::is_zero(pb.z);
    if (negate_b) pb.y = F::select(bz, F::template lc3<-1, 0, 0>(pb.y, pb.y, pb.y), pb.y);
    Jac<F> r = jac_add_flags<F>(pa, pb, F::is_zero(pa.z), bz);
    st(r.x, o); st(r.y, o + W); st(r.z, o + 2 * W);
}
EXPORT void hs_g1_add(const uint32_t *a, const uint32_t *b, int negate_b, uint32_t *o) { hs_add_generic<FqField, 8>(a, b, negate_b, o, ld1, st1); }
EXPORT void hsb_g2_add(const uint32_t *a, const uint32_t *b, int negate_b, uint32_t *o) { hs_add_generic<Fq2Field<F2B>, 16>(a, b, negate_b, o, ld2b, st2b); }
// ---------------------------------------------------------------- four lanes per pairing (quad.hpp) on a simulated quad
typedef Fq2B<FeQ> F2Q;
EXPORT void hsq_fq12_sqr(const uint32_t *a, uint32_t *o) { q12_store(q12_sqr(q12_load<F2Q>(a)), o); }
EXPORT void hsq_fq12_mul(const uint32_t *a, const uint32_t *b, int conj_b, uint32_t *o) { q12_store(q12_mul_half(q12_load<F2Q>(a), q12_load<F2Q>(b).h, conj_b != 0), o); }
EXPORT void hsq_fq12_cyclotomic_sqr(const uint32_t *a, uint32_t *o) { q12_store(q12_cyclotomic_sqr(q12_load<F2Q>(a)), o); }
EXPORT void hsq_fq12_inverse(const uint32_t *a, uint32_t *o) { q12_store(q12_inverse(q12_load<F2Q>(a)), o); }
EXPORT void hsq_fq12_conj(const uint32_t *a, uint32_t *o) { q12_store(q12_conj(q12_load<F2Q>(a)), o); }
EXPORT void hsq_fq12_frobenius(const uint32_t *a, int p, uint32_t *o) {
    QFq12<F2Q> f = q12_load<F2Q>(a);
    q12_store(p == 1 ? q12_frobenius<1>(f) : p == 2 ? q12_frobenius<2>(f) : q12_frobenius<3>(f), o);
}
EXPORT void hsq_fq12_mul_by_024(const uint32_t *a, const uint32_t *l0, const uint32_t *lvw, const uint32_t *lvv, uint32_t *o) {
    q12_store(q12_mul_by_024(q12_load<F2Q>(a), f2_load((F2Q *)0, l0), f2_load((F2Q *)0, lvw), f2_load((F2Q *)0, lvv)), o);
}
EXPORT void hsq_final_exponentiation(const uint32_t *a, uint32_t *o) {
    QuadTableVars<F2Q> tbl;
    q12_store(q_final_exponentiation(q12_load<F2Q>(a), tbl), o);
}
// the whole pairing as the quad kernels run it: prologue on both pairs, NAF Miller loop, final exponentiation
EXPORT void hsq_pairing(const uint32_t *g1, const uint32_t *g2, int final_exp, uint32_t *o) {
    bool inf = words_all_zero(g1 + 16, 8) || words_all_zero(g2 + 32, 16);
    G1Aff<FeQ> p; G2Aff<F2Q> q;
    pair_prologue<FeQ>(f2_scalar_load((F2Q *)0, g1), f2_scalar_load((F2Q *)0, g1 + 8), f2_scalar_load((F2Q *)0, g1 + 16),
                       f2_load((F2Q *)0, g2), f2_load((F2Q *)0, g2 + 16), f2_load((F2Q *)0, g2 + 32), p, q);
    MillerStateVars<F2Q, FeQ> st;
    QFq12<F2Q> f = q_miller_loop_naf(p, q, st);
    if (final_exp) { QuadTableVars<F2Q> tbl; f = q_final_exponentiation(f, tbl); }
    if (inf) f = q12_one<F2Q>();
    q12_store(f, o);
}
// pairing through the NAF Miller schedule (what the pairing kernels run): only the exponentiated value is comparable
EXPORT void hsb_pairing_naf(const uint32_t *g1, const uint32_t *g2, uint32_t *o) {
    bool inf = words_all_zero(g1 + 16, 8) || words_all_zero(g2 + 32, 16);
    G1Aff<FeP> p; G2Aff<F2B> q;
    pair_prologue<FeP>(f2_scalar_load((F2B *)0, g1), f2_scalar_load((F2B *)0, g1 + 8), f2_scalar_load((F2B *)0, g1 + 16),
                       f2_load((F2B *)0, g2), f2_load((F2B *)0, g2 + 16), f2_load((F2B *)0, g2 + 32), p, q);
    MillerStateVars<F2B, FeP> st;
    Fq12<F2B> f = final_exponentiation(miller_loop_sched<true>(p, q, st));
    if (inf) f = f12_one<F2B>();
    f12_store(f, o);
}
// Gt::pow through the windowed chain of bn254_gt_pow_B (lane-pair mapping)
EXPORT void hsb_gt_pow(const uint32_t *a, const uint32_t *k, uint32_t *o) {
    uint32_t raw[8];
    fr_from_mont(k, raw);
    PowTableVars<F2B> tbl;
    f12_store(gt_pow_windowed(f12_load<F2B>(a), raw, tbl), o);
}
EXPORT void hsb_miller(const uint32_t *g1, const uint32_t *g2, uint32_t *o) {
    G1Aff<FeP> p; G2Aff<F2B> q;
    pair_prologue<FeP>(f2_scalar_load((F2B *)0, g1), f2_scalar_load((F2B *)0, g1 + 8), f2_scalar_load((F2B *)0, g1 + 16),
                       f2_load((F2B *)0, g2), f2_load((F2B *)0, g2 + 16), f2_load((F2B *)0, g2 + 32), p, q);
    f12_store(miller_loop(p, q), o);
}


// prepared-G2 mode as the kernels run it: precompute_lines -> 102 coefficients (stored in the reference image) -> miller_loop_prepared
EXPORT void hsb_prepared_pairing(const uint32_t *g1, const uint32_t *g2, uint32_t *coeffs_out, uint32_t *o) {
    G2Aff<F2B> q = g2_to_affine(f2_load((F2B *)0, g2), f2_load((F2B *)0, g2 + 16), f2_load((F2B *)0, g2 + 32));
    auto sink = [&](int idx, const Line<F2B> &l) {
        uint32_t *c = coeffs_out + idx * 48;
        f2_store(l.ell_0, c); f2_store(l.ell_vw, c + 16); f2_store(l.ell_vv, c + 32);
    };
    precompute_lines(q, sink);
    FeP zi = fe_inverse(f2_scalar_load((F2B *)0, g1 + 16)), zi2 = fe_sqr(zi);
    G1Aff<FeP> p = {fe_mul(f2_scalar_load((F2B *)0, g1), zi2), fe_mul(f2_scalar_load((F2B *)0, g1 + 8), fe_mul(zi2, zi))};
    struct PStore { G1Aff<FeP> p_; G1Aff<FeP> get_p() const { return p_; } } ps = {p};
    auto source = [&](int idx) {
        const uint32_t *c = coeffs_out + idx * 48;
        Line<F2B> l = {f2_load((F2B *)0, c), f2_load((F2B *)0, c + 16), f2_load((F2B *)0, c + 32)};
        return l;
    };
    f12_store(final_exponentiation(miller_loop_prepared<F2B>(ps, source)), o);
}
// NATIVE prepared-G2 mode as the kernels run it (pairing.hpp precompute_native -> table -> miller_loop_native -> final exponentiation).
// table: [88 lines][2 lanes][48 words] in the kernel's record layout: A own (9), xi B u (9), xi B v (9), pad, B u (9), B v (9), pad 2
struct NativeTableSim {
    uint32_t *t;                         // [line][lane][48]
    FeP raw[NATIVE_LINES][4];
    static void put(uint32_t *w, const Fe &x) { for (int i = 0; i < 9; ++i) w[i] = x.l[i]; }
    static Fe get(const uint32_t *w) { Fe x; for (int i = 0; i < 9; ++i) x.l[i] = w[i]; BN_SETB(x, 1, 1); return x; }
    void put_raw(int i, const F2B &e0, const F2B &d, const F2B &c, const F2B &pref) { raw[i][0] = e0.v; raw[i][1] = d.v; raw[i][2] = c.v; raw[i][3] = pref.v; }
    void get_raw(int i, F2B &e0, F2B &d, F2B &c) const { e0.v = raw[i][0]; d.v = raw[i][1]; c.v = raw[i][2]; }
    F2B get_prefix(int i) const { return {raw[i][3]}; }
    void put_final(int i, const FeP &a, const Fq2BPrep<FeP> &b, const Fq2BPrep<FeP> &xb) {
        for (int lane = 0; lane < 2; ++lane) {
            uint32_t *w = t + (i * 2 + lane) * 48;
            for (int j = 0; j < 48; ++j) w[j] = 0;
            put(w, a.v[lane]); put(w + 9, xb.u.v[lane]); put(w + 18, xb.v.v[lane]); put(w + 28, b.u.v[lane]); put(w + 37, b.v.v[lane]);
        }
    }
};
struct NativeLineSim {
    const uint32_t *t;
    PNative<FeP> p;
    int line = 0;
    void set_line(int i) { line = i; }
    FeP ld(int off) const { return {{NativeTableSim::get(t + (line * 2 + 0) * 48 + off), NativeTableSim::get(t + (line * 2 + 1) * 48 + off)}}; }
    Fq2BPrep<FeP> x0() const { return f2b_prepare(F2B{fe_mul(ld(0), p.sigma)}); }
    Fq2BPrep<FeP> xb() const { return {ld(9), ld(18)}; }
    Fq2BPrep<FeP> b() const { return {ld(28), ld(37)}; }
    FeP tau() const { return p.tau; }
    FeP tau9() const { return p.tau9; }
    FeP taum() const { return p.taum; }
};
EXPORT int hsb_native_lines() { return NATIVE_LINES; }
EXPORT void hsb_native_precompute(const uint32_t *g2, uint32_t *table) {
    G2Aff<F2B> q = g2_to_affine(f2_load((F2B *)0, g2), f2_load((F2B *)0, g2 + 16), f2_load((F2B *)0, g2 + 32));
    static NativeTableSim st;
    st.t = table;
    precompute_native(q, st);
}
// miller: 1 = the un-exponentiated Miller value (differs from the reference's by subfield factors), 0 = the pairing
EXPORT void hsb_native_pairing(const uint32_t *g1, const uint32_t *table, int miller, uint32_t *o) {
    const bool inf = words_all_zero(g1 + 16, 8);
    NativeLineSim src;
    src.t = table;
    src.p = p_native(f2_scalar_load((F2B *)0, g1), f2_scalar_load((F2B *)0, g1 + 8), f2_scalar_load((F2B *)0, g1 + 16));
    Fq12<F2B> f = miller_loop_native<F2B>(src);
    if (!miller) f = final_exponentiation(f);
    if (inf) f = f12_one<F2B>();
    f12_store(f, o);
}
// the multi-pairing over native tables (pairing.hpp miller_loop_native_shared): m pairs on one simulated lane pair share the accumulator; a pair
// with a point at infinity (flags[i] != 0, or z_P = 0) reads the identity record with sigma = 1, tau = 0.  tables: m x [88][2][48] words.
struct NativeSharedSim {
    const uint32_t *t[8];
    FeP sigma[8], tau_[8];
    uint32_t ident[2][48];
    int line = 0, cur = 0;
    void set_line(int l, int i) { line = l; cur = i; }
    FeP ld(int off) const {
        const uint32_t *base = t[cur];
        if (!base) return {{NativeTableSim::get(ident[0] + off), NativeTableSim::get(ident[1] + off)}};
        return {{NativeTableSim::get(base + (line * 2 + 0) * 48 + off), NativeTableSim::get(base + (line * 2 + 1) * 48 + off)}};
    }
    Fq2BPrep<FeP> x0() const { return f2b_prepare(F2B{fe_mul(ld(0), sigma[cur])}); }
    Fq2BPrep<FeP> xb() const { return {ld(9), ld(18)}; }
    Fq2BPrep<FeP> b() const { return {ld(28), ld(37)}; }
    FeP tau() const { return tau_[cur]; }
    FeP tau9() const { return p_native_tau9(tau_[cur]); }             // re-derived per line, like the kernel (LDS holds sigma, tau only)
    FeP taum() const { return p_native_taum(tau_[cur]); }
};
// miller_only != 0: the un-exponentiated value (what the kernel alone executes)
EXPORT void hsb_native_product(int m, const uint32_t *g1, const uint32_t *tables, const uint32_t *q_inf, int miller_only, uint32_t *o) {
    static NativeSharedSim src;
    for (int j = 0; j < 48; ++j) src.ident[0][j] = src.ident[1][j] = 0;
    for (int j = 0; j < 9; ++j) src.ident[0][j] = k::ONE[j];                        // A = 1 + 0 i: even lane ONE, odd lane 0; B = xi B = 0
    for (int i = 0; i < m; ++i) {
        const uint32_t *w = g1 + 24 * i;
        const bool inf = words_all_zero(w + 16, 8) || q_inf[i] != 0;
        src.t[i] = inf ? nullptr : tables + (size_t)i * NATIVE_LINES * 2 * 48;
        const PNative<FeP> pn = p_native(f2_scalar_load((F2B *)0, w), f2_scalar_load((F2B *)0, w + 8), f2_scalar_load((F2B *)0, w + 16));
        src.sigma[i] = pn.sigma; src.tau_[i] = pn.tau;
        if (inf) p_native_identity(src.sigma[i], src.tau_[i]);
    }
    Fq12<F2B> f = m == 1 ? miller_loop_native_shared<1, F2B>(src) : m == 2 ? miller_loop_native_shared<2, F2B>(src) : m == 3 ? miller_loop_native_shared<3, F2B>(src) : miller_loop_native_shared<4, F2B>(src);
    f12_store(miller_only ? f : final_exponentiation(f), o);
}
// bn254_miller_prepared_B alone (the reference-image coefficients already computed): what that kernel executes per pairing
EXPORT void hsb_prepared_miller(const uint32_t *g1, const uint32_t *coeffs, uint32_t *o) {
    FeP zi = fe_inverse(f2_scalar_load((F2B *)0, g1 + 16)), zi2 = fe_sqr(zi);
    G1Aff<FeP> p = {fe_mul(f2_scalar_load((F2B *)0, g1), zi2), fe_mul(f2_scalar_load((F2B *)0, g1 + 8), fe_mul(zi2, zi))};
    struct PStore { G1Aff<FeP> p_; G1Aff<FeP> get_p() const { return p_; } } ps = {p};
    auto source = [&](int idx) {
        const uint32_t *c = coeffs + idx * 48;
        Line<F2B> l = {f2_load((F2B *)0, c), f2_load((F2B *)0, c + 16), f2_load((F2B *)0, c + 32)};
        return l;
    };
    f12_store(miller_loop_prepared<F2B>(ps, source), o);
}
// the product tree step of the multi-pairing: acc = acc * x repeatedly (bn254_gt_product_B)
EXPORT void hsb_gt_product(const uint32_t *in, int n, uint32_t *o) {
    Fq12<F2B> acc = f12_load<F2B>(in);
    for (int j = 1; j < n; ++j) acc = f12_mul_o(acc, f12_load<F2B>(in + 96 * j));
    f12_store(acc, o);
}
#ifdef BN_BOUNDS
EXPORT void hs_counts_reset() { op_counts() = OpCounts{}; }
EXPORT void hs_counts_get(unsigned long *o) { OpCounts c = op_counts(); o[0] = c.mul; o[1] = c.mul2; o[2] = c.lc3; o[3] = c.lc3w; o[4] = c.norm; o[5] = c.addsub; o[6] = c.reduce; o[7] = c.select; o[8] = c.macs; }
// the fused NAF Miller loop alone, as bn254_miller_naf_B runs it (prologue included): the executed-chain count of the headline kernel
EXPORT void hsb_miller_naf(const uint32_t *g1, const uint32_t *g2, uint32_t *o) {
    G1Aff<FeP> p; G2Aff<F2B> q;
    pair_prologue<FeP>(f2_scalar_load((F2B *)0, g1), f2_scalar_load((F2B *)0, g1 + 8), f2_scalar_load((F2B *)0, g1 + 16),
                       f2_load((F2B *)0, g2), f2_load((F2B *)0, g2 + 16), f2_load((F2B *)0, g2 + 32), p, q);
    MillerStateVars<F2B, FeP> st;
    f12_store(miller_loop_sched<true>(p, q, st), o);
}
EXPORT void hsb_miller_only(const uint32_t *g1, const uint32_t *g2, uint32_t *o) {
    G1Aff<FeP> p; G2Aff<F2B> q;
    pair_prologue<FeP>(f2_scalar_load((F2B *)0, g1), f2_scalar_load((F2B *)0, g1 + 8), f2_scalar_load((F2B *)0, g1 + 16),
                       f2_load((F2B *)0, g2), f2_load((F2B *)0, g2 + 16), f2_load((F2B *)0, g2 + 32), p, q);
    f12_store(miller_loop(p, q), o);
}
#endif

// wire format through the engine code
EXPORT void hs_g1_encode(const uint32_t *p, uint8_t *o) { g1_encode_record(p, o); }
EXPORT void hs_g2_encode(const uint32_t *p, uint8_t *o) { g2_encode_record(p, o); }
EXPORT void hs_fr_encode(const uint32_t *k, uint8_t *out) { fr_encode_record(k, out); }
EXPORT int hs_fr_decode(const uint8_t *in, uint32_t *out) { return fr_decode_record(in, out); }
EXPORT int hs_g1_decode(const uint8_t *in, uint32_t *o) { return g1_decode_record(in, o); }
EXPORT int hs_g2_decode(const uint8_t *in, uint32_t *o) { return g2_decode_record(in, o); }

// ---------------------------------------------------------------- the wave-cooperative Fq12 machine (bn_amd/csrc/wave.hpp)
#include "wavesim.hpp"
// which: 0 MUL (res * slot0), 1 MULC (res * conj(slot0)), 2 CYC, 3..5 FROB1..3, 6 EASY, 7 HARD, 8 FE, 9 CYC5 (five squarings as one
// fused run).   `b` is only read by 0, 1.
EXPORT void hsw_run(int which, const uint32_t *a, const uint32_t *b, uint32_t *o) {
    using namespace bn254::wv;
    static WaveSimShared sh;
    wavesim_init(sh);
    const uint32_t *progs[] = {PROG_MUL, PROG_MULC, PROG_CYC, PROG_FROB1, PROG_FROB2, PROG_FROB3, PROG_EASY, PROG_HARD, PROG_FE, PROG_CYC5};
    const uint32_t *prog = progs[which];
    wavesim_run(sh, [&](WaveSim &w) {
        if (which <= 1) {
            w_load_f12(w, b, OFF_RES); w.sync();
            w_run(w, PROG_PUT0);
        }
        w_load_f12(w, a, OFF_RES); w.sync();
        w_run(w, prog);
        w_store_f12(w, OFF_RES, o);
    });
}

// Gt::pow on the cyclotomic subgroup (bn254_gt_pow_B's fast path) and the membership test that selects it
EXPORT void hsb_gt_pow_cyclotomic(const uint32_t *a, const uint32_t *k, uint32_t *o) {
    uint32_t raw[8];
    fr_from_mont(k, raw);
    PowTableVars<F2B> tbl;
    f12_store(gt_pow_cyclotomic(f12_load<F2B>(a), raw, tbl), o);
}
// Gt::pow through the Frobenius decomposition (the default path of bn254_gt_pow_B for cyclotomic input)
EXPORT void hsb_gt_pow_gls(const uint32_t *a, const uint32_t *k, uint32_t *o) {
    uint32_t raw[8];
    fr_from_mont(k, raw);
    PowTableVars<F2B> tbl;
    f12_store(gt_pow_gls(f12_load<F2B>(a), raw, tbl), o);
}
EXPORT int hsb_gt_is_cyclotomic(const uint32_t *a) { return gt_is_cyclotomic(f12_load<F2B>(a)) ? 1 : 0; }
// what bn254_gt_pow_B executes for one element: membership test, then the chain it selects
EXPORT void hsb_gt_pow_auto(const uint32_t *a, const uint32_t *k, uint32_t *o) {
    uint32_t raw[8];
    fr_from_mont(k, raw);
    PowTableVars<F2B> tbl;
    const Fq12<F2B> base = f12_load<F2B>(a);
    f12_store(gt_is_cyclotomic(base) ? gt_pow_gls(base, raw, tbl) : gt_pow_windowed(base, raw, tbl), o);
}

// G2 * Fr through the GLS chain (what bn254_g2_mul_batch runs), normalized; and the 4-dimensional decomposition itself
template <class F2X>
static void hs_g2_gls_generic(const uint32_t *pt, const uint32_t *k, uint32_t *o) {
    typedef Fq2Field<F2X> F;
    Jac<F> p = {f2_load((F2X *)0, pt), f2_load((F2X *)0, pt + 16), f2_load((F2X *)0, pt + 32)};
    uint32_t raw[8];
    fr_from_mont(k, raw);
    Jac<F> r = jac_normalize<F>(scalar_mul_gls<F2X>(p, raw));
    f2_store(r.x, o); f2_store(r.y, o + 16); f2_store(r.z, o + 32);
}
EXPORT void hs_g2_mul_gls(const uint32_t *pt, const uint32_t *k, uint32_t *o) { hs_g2_gls_generic<F2>(pt, k, o); }
EXPORT void hsb_g2_mul_gls(const uint32_t *pt, const uint32_t *k, uint32_t *o) { hs_g2_gls_generic<F2B>(pt, k, o); }
EXPORT void hs_gls_decompose(const uint32_t *k, uint32_t *o) {         // o: 4 x (3 words magnitude, 1 word sign)
    uint32_t raw[8];
    fr_from_mont(k, raw);
    GlsSplit g = gls_decompose(raw);
    for (int i = 0; i < 4; ++i) { o[4 * i] = g.m[i][0]; o[4 * i + 1] = g.m[i][1]; o[4 * i + 2] = g.m[i][2]; o[4 * i + 3] = g.neg[i]; }
}

// the Miller loop and the whole pairing on the wave machine: Jacobian G1 / G2 images in, Fq12 image out (infinity -> one)
EXPORT void hsw_pairing(int fe, const uint32_t *g1, const uint32_t *g2, uint32_t *o) {
    using namespace bn254::wv;
    static WaveSimShared sh;
    wavesim_init(sh);
    const bool inf = words_all_zero(g1 + 16, 8) || words_all_zero(g2 + 32, 16);
    wavesim_run(sh, [&](WaveSim &w) {
        w_load_points(w, g1, g2); w.sync();
        w_run(w, fe ? PROG_PAIRING : PROG_MILLER);
        if (inf) { w_set_one(w); w.sync(); }
        w_store_f12(w, OFF_RES, o);
    });
}

// the multi-pairing's Miller loop with a shared accumulator (pairing.hpp miller_loop_shared, bn254_miller_shared{2,4}_B):
// FE(shared Miller value of m pairs) must equal the product of the m pairings; pairs with z = 0 contribute 1
template <int M>
struct MillerStateVarsM {
    G2Proj<F2B> r[M]; G2Aff<F2B> b[M]; G1Aff<FeP> p[M]; bool inf[M];
    void put_r(int i, const G2Proj<F2B> &v) { r[i] = v; }
    G2Proj<F2B> get_r(int i) const { return r[i]; }
    void put_base(int i, const G2Aff<F2B> &v) { b[i] = v; }
    G2Aff<F2B> get_base(int i) const { return b[i]; }
    G1Aff<FeP> get_p(int i) const { return p[i]; }
    bool is_inf(int i) const { return inf[i]; }
};
template <int M>
static void hsb_shared(const uint32_t *g1, const uint32_t *g2, uint32_t *o) {
    MillerStateVarsM<M> st;
    for (int i = 0; i < M; ++i) {
        const uint32_t *w1 = g1 + 24 * i, *w2 = g2 + 48 * i;
        st.inf[i] = words_all_zero(w1 + 16, 8) || words_all_zero(w2 + 32, 16);
        G1Aff<FeP> p; G2Aff<F2B> q;
        pair_prologue<FeP>(f2_scalar_load((F2B *)0, w1), f2_scalar_load((F2B *)0, w1 + 8), f2_scalar_load((F2B *)0, w1 + 16),
                           f2_load((F2B *)0, w2), f2_load((F2B *)0, w2 + 16), f2_load((F2B *)0, w2 + 32), p, q);
        const FeP t2 = f2_scalar_const((F2B *)0, k::ISO_T2), t3 = f2_scalar_const((F2B *)0, k::ISO_T3);
        p = {fe_mul(p.x, t2), fe_mul(p.y, t3)};
        q = {f2_scale(q.x, t2), f2_scale(q.y, t3)};
        st.p[i] = p; st.b[i] = q; st.r[i] = G2Proj<F2B>{q.x, q.y, f2_one((F2B *)0)};
    }
    f12_store(final_exponentiation(miller_loop_shared<M, F2B, FeP>(st)), o);
}
EXPORT void hsb_pairing_product_shared(int m, const uint32_t *g1, const uint32_t *g2, uint32_t *o) {
    if (m == 2) hsb_shared<2>(g1, g2, o); else hsb_shared<4>(g1, g2, o);
}

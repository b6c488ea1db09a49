// TEST INFRASTRUCTURE (host simulation only): a 4-lane value type - two simulated lane pairs - that executes the four-lanes-per-pairing
// mapping of bn_amd/csrc/quad.hpp on the CPU.  Every primitive of lanepair.hpp is lifted once more; quad_xchg swaps the two pairs (the
// GPU does it with DPP quad_perm [2,3,0,1]), quad_pick takes the first argument on the lower pair and the second on the upper pair.
#pragma once
#include "lanepair.hpp"

namespace bn254 {
struct FeQ { FeP p[2]; };

BN_FN FeQ quad_xchg(const FeQ &x) { return {{x.p[1], x.p[0]}}; }
BN_FN FeQ quad_pick(const FeQ &lower_choice, const FeQ &upper_choice) { return {{lower_choice.p[0], upper_choice.p[1]}}; }
BN_FN FeQ quad_lo(const FeQ &x) { return {{x.p[0], x.p[0]}}; }      // the lower pair's value on both pairs (GPU: DPP quad_perm [0,1,0,1])
BN_FN FeQ quad_up(const FeQ &x) { return {{x.p[1], x.p[1]}}; }      // the upper pair's ([2,3,2,3])
#define BN_Q1(NAME) BN_FN FeQ NAME(const FeQ &a) { return {{NAME(a.p[0]), NAME(a.p[1])}}; }
#define BN_Q2(NAME) BN_FN FeQ NAME(const FeQ &a, const FeQ &b) { return {{NAME(a.p[0], b.p[0]), NAME(a.p[1], b.p[1])}}; }
BN_Q1(lane_partner) BN_Q2(lane_pick) BN_Q2(fe_add) BN_Q1(fe_dbl) BN_Q2(fe_ssub) BN_Q1(fe_norm) BN_Q1(fe_half) BN_Q1(fe_std)
BN_Q2(fe_mul) BN_Q2(fe_mul_body) BN_Q1(fe_sqr) BN_Q1(fe_inverse)
#undef BN_Q1
#undef BN_Q2
BN_FN FeQ lane_bcast(const FeQ *, const Fe &x) { return {{lane_bcast((const FeP *)nullptr, x), lane_bcast((const FeP *)nullptr, x)}}; }
template <class TAB> BN_FN FeQ lane_const_pick(const FeQ *, const TAB &e, const TAB &o) { return {{lane_const_pick((const FeP *)nullptr, e, o), lane_const_pick((const FeP *)nullptr, e, o)}}; }
template <int LB, int K> BN_FN FeQ fe_sub(const FeQ &a, const FeQ &b) { return {{fe_sub<LB, K>(a.p[0], b.p[0]), fe_sub<LB, K>(a.p[1], b.p[1])}}; }
template <int LB, int K> BN_FN FeQ fe_neg(const FeQ &a) { return {{fe_neg<LB, K>(a.p[0]), fe_neg<LB, K>(a.p[1])}}; }
template <int C1, int C2, int C3> BN_FN FeQ fe_lc3w(const FeQ &x, const FeQ &y, const FeQ &z) { return {{fe_lc3w<C1, C2, C3>(x.p[0], y.p[0], z.p[0]), fe_lc3w<C1, C2, C3>(x.p[1], y.p[1], z.p[1])}}; }
template <int C1, int C2, int C3> BN_FN FeQ fe_lc3(const FeQ &x, const FeQ &y, const FeQ &z) { return {{fe_lc3<C1, C2, C3>(x.p[0], y.p[0], z.p[0]), fe_lc3<C1, C2, C3>(x.p[1], y.p[1], z.p[1])}}; }
template <int C1, int C2, int C3> BN_FN FeQ fe_lc3_par(const FeQ &x, const FeQ &y, const FeQ &z) { return {{fe_lc3_par<C1, C2, C3>(x.p[0], y.p[0], z.p[0]), fe_lc3_par<C1, C2, C3>(x.p[1], y.p[1], z.p[1])}}; }
template <int C1, int C2, int C3, int C4>
BN_FN FeQ fe_lc4_par(const FeQ &x, const FeQ &y, const FeQ &z, const FeQ &w) { return {{fe_lc4_par<C1, C2, C3, C4>(x.p[0], y.p[0], z.p[0], w.p[0]), fe_lc4_par<C1, C2, C3, C4>(x.p[1], y.p[1], z.p[1], w.p[1])}}; }
template <int C1, int C2, int C3, int C4>
BN_FN FeQ fe_lc4w_par(const FeQ &x, const FeQ &y, const FeQ &z, const FeQ &w) { return {{fe_lc4w_par<C1, C2, C3, C4>(x.p[0], y.p[0], z.p[0], w.p[0]), fe_lc4w_par<C1, C2, C3, C4>(x.p[1], y.p[1], z.p[1], w.p[1])}}; }
template <int C1, int C2, int C3> BN_FN FeQ fe_lc3sw(const FeQ &x, const FeQ &y, const FeQ &z) { return {{fe_lc3sw<C1, C2, C3>(x.p[0], y.p[0], z.p[0]), fe_lc3sw<C1, C2, C3>(x.p[1], y.p[1], z.p[1])}}; }
BN_FN FeQ fe_sdiff(const FeQ &a, const FeQ &b) { return {{fe_sdiff(a.p[0], b.p[0]), fe_sdiff(a.p[1], b.p[1])}}; }
BN_FN FeQ fe_sneg(const FeQ &a) { return {{fe_sneg(a.p[0]), fe_sneg(a.p[1])}}; }
BN_FN FeQ fe_mul2s(const FeQ &a, const FeQ &u, const FeQ &c, const FeQ &v) { return {{fe_mul2s(a.p[0], u.p[0], c.p[0], v.p[0]), fe_mul2s(a.p[1], u.p[1], c.p[1], v.p[1])}}; }
BN_FN FeQ fe_mul2(const FeQ &a, const FeQ &u, const FeQ &c, const FeQ &v) { return {{fe_mul2(a.p[0], u.p[0], c.p[0], v.p[0]), fe_mul2(a.p[1], u.p[1], c.p[1], v.p[1])}}; }
BN_FN FeQ fe_mul6(const FeQ &a1, const FeQ &u1, const FeQ &c1, const FeQ &v1, const FeQ &a2, const FeQ &u2, const FeQ &c2, const FeQ &v2,
                  const FeQ &a3, const FeQ &u3, const FeQ &c3, const FeQ &v3) {
    return {{fe_mul6(a1.p[0], u1.p[0], c1.p[0], v1.p[0], a2.p[0], u2.p[0], c2.p[0], v2.p[0], a3.p[0], u3.p[0], c3.p[0], v3.p[0]),
             fe_mul6(a1.p[1], u1.p[1], c1.p[1], v1.p[1], a2.p[1], u2.p[1], c2.p[1], v2.p[1], a3.p[1], u3.p[1], c3.p[1], v3.p[1])}};
}
BN_FN FeQ fe_select(bool take_b, const FeQ &a, const FeQ &b) { return {{fe_select(take_b, a.p[0], b.p[0]), fe_select(take_b, a.p[1], b.p[1])}}; }
// both pairs of a quad read the same inputs (G1 / G2 points) ...
BN_FN FeQ lane_load_pair(const FeQ *, const uint32_t *w0, const uint32_t *w1) { return {{lane_load_pair((const FeP *)nullptr, w0, w1), lane_load_pair((const FeP *)nullptr, w0, w1)}}; }
BN_FN bool lane_pair_all_zero(const FeQ &a) { return lane_pair_all_zero(a.p[0]); }
BN_FN bool lane_pair_all_zero_std(const FeQ &a) { return lane_pair_all_zero_std(a.p[0]); }
}  // namespace bn254
#include "../../bn_amd/csrc/fq2.hpp"
namespace bn254 {
// ... and each its own half of an Fq12: the lower pair the Fq2 at w, the upper pair the one at w + off
BN_FN Fq2B<FeQ> quad_load_f2(const Fq2B<FeQ> *, const uint32_t *w, int off) {
    return {{{lane_load_pair((const FeP *)nullptr, w, w + 8), lane_load_pair((const FeP *)nullptr, w + off, w + off + 8)}}};
}
BN_FN void quad_store_f2(const Fq2B<FeQ> &a, uint32_t *w, int off) { lane_store_pair(a.v.p[0], w, w + 8); lane_store_pair(a.v.p[1], w + off, w + off + 8); }
}  // namespace bn254

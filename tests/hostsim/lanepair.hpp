// TEST INFRASTRUCTURE (host simulation only): a 2-lane value type that executes the lane-pair Fq2 mapping (Fq2B) on the CPU.
// Every fe_* primitive is lifted element-wise; lane_partner swaps the lanes (the GPU does it with DPP quad_perm [1,0,3,2]).
#pragma once
#include "../../bn_amd/csrc/fe.hpp"

namespace bn254 {
struct FeP { Fe v[2]; };

BN_FN FeP lane_partner(const FeP &x) { return {{x.v[1], x.v[0]}}; }
BN_FN FeP lane_pick(const FeP &even_choice, const FeP &odd_choice) { return {{even_choice.v[0], odd_choice.v[1]}}; }
BN_FN FeP lane_bcast(const FeP *, const Fe &x) { return {{x, x}}; }
template <class TAB> BN_FN FeP lane_const_pick(const FeP *, const TAB &even_tab, const TAB &odd_tab) { return {{fe_const(even_tab), fe_const(odd_tab)}}; }
BN_FN FeP fe_add(const FeP &a, const FeP &b) { return {{fe_add(a.v[0], b.v[0]), fe_add(a.v[1], b.v[1])}}; }
BN_FN FeP fe_dbl(const FeP &a) { return fe_add(a, a); }
template <int LB, int K> BN_FN FeP fe_sub(const FeP &a, const FeP &b) { return {{fe_sub<LB, K>(a.v[0], b.v[0]), fe_sub<LB, K>(a.v[1], b.v[1])}}; }
template <int LB, int K> BN_FN FeP fe_neg(const FeP &a) { return {{fe_neg<LB, K>(a.v[0]), fe_neg<LB, K>(a.v[1])}}; }
BN_FN FeP fe_ssub(const FeP &a, const FeP &b) { return {{fe_ssub(a.v[0], b.v[0]), fe_ssub(a.v[1], b.v[1])}}; }
BN_FN FeP fe_sdiff(const FeP &a, const FeP &b) { return {{fe_sdiff(a.v[0], b.v[0]), fe_sdiff(a.v[1], b.v[1])}}; }
BN_FN FeP fe_sneg(const FeP &a) { return {{fe_sneg(a.v[0]), fe_sneg(a.v[1])}}; }
BN_FN FeP fe_mul2s(const FeP &a, const FeP &u, const FeP &c, const FeP &v) {
    return {{fe_mul2s(a.v[0], u.v[0], c.v[0], v.v[0]), fe_mul2s(a.v[1], u.v[1], c.v[1], v.v[1])}};
}
template <int C1, int C2, int C3>
BN_FN FeP fe_lc3w(const FeP &x, const FeP &y, const FeP &z) { return {{fe_lc3w<C1, C2, C3>(x.v[0], y.v[0], z.v[0]), fe_lc3w<C1, C2, C3>(x.v[1], y.v[1], z.v[1])}}; }
BN_FN FeP fe_norm(const FeP &a) { return {{fe_norm(a.v[0]), fe_norm(a.v[1])}}; }
BN_FN FeP fe_half(const FeP &a) { return {{fe_half(a.v[0]), fe_half(a.v[1])}}; }
BN_FN FeP fe_std(const FeP &a) { return {{fe_std(a.v[0]), fe_std(a.v[1])}}; }
template <int C1, int C2, int C3>
BN_FN FeP fe_lc3(const FeP &x, const FeP &y, const FeP &z) { return {{fe_lc3<C1, C2, C3>(x.v[0], y.v[0], z.v[0]), fe_lc3<C1, C2, C3>(x.v[1], y.v[1], z.v[1])}}; }
template <int C1, int C2, int C3>      // middle term: minus on the even lane, plus on the odd lane
BN_FN FeP fe_lc3_par(const FeP &x, const FeP &y, const FeP &z) {
    return {{fe_lc3_core<C1, C2, C3, true>(x.v[0], y.v[0], z.v[0], true), fe_lc3_core<C1, C2, C3, true>(x.v[1], y.v[1], z.v[1], false)}};
}
template <int C1, int C2, int C3, int C4>
BN_FN FeP fe_lc4_par(const FeP &x, const FeP &y, const FeP &z, const FeP &w) {
    return {{fe_lc4_core<C1, C2, C3, C4, false, true>(x.v[0], y.v[0], z.v[0], w.v[0], true), fe_lc4_core<C1, C2, C3, C4, false, true>(x.v[1], y.v[1], z.v[1], w.v[1], false)}};
}
template <int C1, int C2, int C3, int C4>
BN_FN FeP fe_lc4w_par(const FeP &x, const FeP &y, const FeP &z, const FeP &w) {
    return {{fe_lc4_core<C1, C2, C3, C4, true>(x.v[0], y.v[0], z.v[0], w.v[0], true), fe_lc4_core<C1, C2, C3, C4, true>(x.v[1], y.v[1], z.v[1], w.v[1], false)}};
}
template <int C1, int C2, int C3>
BN_FN FeP fe_lc3sw(const FeP &x, const FeP &y, const FeP &z) {
    return {{fe_lc4_core<C1, C2, C3, 0, true>(x.v[0], y.v[0], z.v[0], z.v[0], false), fe_lc4_core<C1, C2, C3, 0, true>(x.v[1], y.v[1], z.v[1], z.v[1], false)}};
}
BN_FN FeP fe_mul(const FeP &a, const FeP &b) { return {{fe_mul(a.v[0], b.v[0]), fe_mul(a.v[1], b.v[1])}}; }
BN_FN FeP fe_mul_body(const FeP &a, const FeP &b) { return fe_mul(a, b); }
BN_FN FeP fe_sqr(const FeP &a) { return fe_mul(a, a); }
BN_FN FeP fe_mul2(const FeP &a, const FeP &u, const FeP &c, const FeP &v) {
    return {{fe_mul2(a.v[0], u.v[0], c.v[0], v.v[0]), fe_mul2(a.v[1], u.v[1], c.v[1], v.v[1])}};
}
BN_FN FeP fe_mul6(const FeP &a1, const FeP &u1, const FeP &c1, const FeP &v1, const FeP &a2, const FeP &u2, const FeP &c2, const FeP &v2,
                  const FeP &a3, const FeP &u3, const FeP &c3, const FeP &v3) {
    return {{fe_mul6(a1.v[0], u1.v[0], c1.v[0], v1.v[0], a2.v[0], u2.v[0], c2.v[0], v2.v[0], a3.v[0], u3.v[0], c3.v[0], v3.v[0]),
             fe_mul6(a1.v[1], u1.v[1], c1.v[1], v1.v[1], a2.v[1], u2.v[1], c2.v[1], v2.v[1], a3.v[1], u3.v[1], c3.v[1], v3.v[1])}};
}
BN_FN FeP fe_mul5(const FeP &a1, const FeP &u1, const FeP &c1, const FeP &v1, const FeP &a2, const FeP &u2, const FeP &c2, const FeP &v2, const FeP &a3, const FeP &u3) {
    return {{fe_mul5(a1.v[0], u1.v[0], c1.v[0], v1.v[0], a2.v[0], u2.v[0], c2.v[0], v2.v[0], a3.v[0], u3.v[0]),
             fe_mul5(a1.v[1], u1.v[1], c1.v[1], v1.v[1], a2.v[1], u2.v[1], c2.v[1], v2.v[1], a3.v[1], u3.v[1])}};
}
BN_FN FeP fe_canonical(const FeP &a) { return {{fe_canonical(a.v[0]), fe_canonical(a.v[1])}}; }
struct Fe;
BN_FN Fe fe_cneg(bool flag, const Fe &z);         // wave.hpp
BN_FN FeP fe_cneg(bool flag, const FeP &a) { return {{fe_cneg(flag, a.v[0]), fe_cneg(flag, a.v[1])}}; }
BN_FN FeP fe_inverse(const FeP &a) { return {{fe_inverse(a.v[0]), fe_inverse(a.v[1])}}; }
BN_FN FeP fe_select(bool take_b, const FeP &a, const FeP &b) { return {{fe_select(take_b, a.v[0], b.v[0]), fe_select(take_b, a.v[1], b.v[1])}}; }
// I/O of a pair: the even lane reads/writes w0, the odd lane w1
BN_FN FeP lane_load_pair(const FeP *, const uint32_t *w0, const uint32_t *w1) { return {{fe_from_u32x8(w0), fe_from_u32x8(w1)}}; }
BN_FN void lane_store_pair(const FeP &a, uint32_t *w0, uint32_t *w1) { fe_to_u32x8(a.v[0], w0); fe_to_u32x8(a.v[1], w1); }
BN_FN bool lane_pair_all_zero(const FeP &a) { return fe_is_zero(a.v[0]) && fe_is_zero(a.v[1]); }
BN_FN bool lane_pair_all_zero_std(const FeP &a) { return fe_is_zero_std(a.v[0]) && fe_is_zero_std(a.v[1]); }
}  // namespace bn254

// The WAVE-COOPERATIVE Fq12 machine: one Fq12 operation spread over the 32 lane pairs of a wave, for the latency-bound tails of
// the path - the single final exponentiation of a multi-pairing (fq12.rs:41-88 after the fold of shootout/main.rs:11-16), the last
// levels of its Fq12 product tree, and batches too small to fill the chip.  In the lane-pair kernels (bn254_kernels_b.hip) a
// final exponentiation is a 2.0 ms serial chain on TWO lanes whatever the batch size; here every Fq2 product of an Fq12 product
// (18, Karatsuba) or of a Granger-Scott squaring (9 Fq2 squarings) runs on its own lane pair at the same time.
//
// Data: a register file of Fq2 values in LDS, [page][limb][slot]: a register = two adjacent slots (even lane: c0, odd lane: c1),
// 64 slots per page, so the 64 lanes of a wave reading 64 different slots of one limb row hit 64 different banks.
// Control: programs of table-driven phases, generated AND verified on exact field elements by tools/gen_wave_tables.py:
//     PROD   R[dst] = (sum of up to 4 registers, optionally conjugated) * (sum of up to 4 registers), or the square of the first
//     COMB   R[dst] = reduce(CX xi (R[x0] - R[x1] - R[x2]) + CY (R[y0] + R[y1] - R[y2] - R[y3]) + zs CZ R[z])
//     INV    R[dst] = 1 / (gathered value)       (fq2.rs:125-136; one constant-time divsteps inversion)
// Each lane pair reads its role (which registers, which destination) from a table: the instruction stream is lane-uniform, every
// branch is wave-uniform.  The leaves are the lane-pair leaves of fq2.hpp (f2b_mul_body / f2b_sqr_body / fe_lc4_core), so the
// number system, its bounds and the bytes that leave the engine are those of the batch kernels.
//
// W (the wave context) provides: T (Fe on the GPU; the host simulation's 2-lane value), ld(byte offset) / st(byte offset, T) on
// the register file, role(phase) = this pair's role, odd-lane flags via the lane primitives of fq2.hpp, and sync().
#pragma once
#include "fq2.hpp"
#include "wave_tables.hpp"

namespace bn254 {
// flag ? -z : z as a signed lazy value (consumed by the fused reductions only)
BN_FN Fe fe_cneg(bool flag, const Fe &z) {
    BN_REQUIRE(!z.sg, "fe_cneg input");
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = flag ? 0u - z.l[i] : z.l[i];
    BN_SETB(r, z.lb, z.vb);
    BN_IFB(r.sg = true;)
    return r;
}
namespace wv {

// REL = true: the table may hold indices relative to the program entry's base register (products by a table slot, slot copies)
template <bool RELX>
BN_FN uint32_t w_addr(uint32_t idx, uint32_t base) {
    if constexpr (RELX) return (idx & REL) ? (idx & 0x7fffu) + base : idx;
    else return idx;
}

template <int NA, int NB, bool SQR, bool CONJ, bool RELX, class W>
BN_FN void w_prod(W &w, const Role &r, uint32_t base) {
    using T = typename W::T;
    T a = w.ld(w_addr<RELX>(r.src[0], base));
#pragma unroll
    for (int k = 1; k < NA; ++k) a = fe_add(a, w.ld(w_addr<RELX>(r.src[k], base)));
    if (NA > 1) a = fe_norm(a);
    if constexpr (CONJ) {                                              // conjugate of the first operand: the odd lane negates (lazy: lb 2, vb 4)
        T n = lane_pick(a, fe_neg<1, 4>(a));
        a = fe_select((r.flags & 2) != 0, a, n);
    }
    T res;
    if constexpr (SQR) {
        res = f2b_sqr_inl(a);
    } else {
        T b = w.ld(w_addr<RELX>(r.src[4], base));
#pragma unroll
        for (int k = 1; k < NB; ++k) b = fe_add(b, w.ld(w_addr<RELX>(r.src[4 + k], base)));
        if (NB > 1) b = fe_norm(b);
        res = f2b_mul_inl(a, b);
    }
    if (r.flags & 1) w.st(w_addr<RELX>(r.dst, base), res);
}

// the reduction of one output coefficient: x = R[x0] - R[x1] - R[x2], y = R[y0] + R[y1] - R[y2] - R[y3] as signed lazy limb sums,
//   M: xi x + y              = fe_lc3_par<9, 1, 1>(x, partner x, y)             (f2_lc_xi<1, 1> of fq2.hpp)
//   C: 3 xi x + 3 y +- 2 z   = fe_lc4_par<27, 3, 3, 2>(x, partner x, y, +-z)    (f2_lc_xi2 of fq2.hpp; the Granger-Scott update)
template <bool CYC, bool RELX, class W>
BN_FN void w_comb(W &w, const Role &r, uint32_t base) {
    using T = typename W::T;
    T x = fe_ssub(fe_ssub(w.ld(w_addr<RELX>(r.src[0], base)), w.ld(w_addr<RELX>(r.src[1], base))), w.ld(w_addr<RELX>(r.src[2], base)));
    T y = fe_ssub(fe_ssub(fe_add(w.ld(w_addr<RELX>(r.src[3], base)), w.ld(w_addr<RELX>(r.src[4], base))), w.ld(w_addr<RELX>(r.src[5], base))), w.ld(w_addr<RELX>(r.src[6], base)));
    T res;
    if constexpr (CYC) {
        T z = fe_cneg((r.flags & 2) != 0, w.ld(w_addr<RELX>(r.src[7], base)));
        res = fe_lc4_par<27, 3, 3, 2>(x, lane_partner(x), y, z);
    } else {
        res = fe_lc4_par<9, 1, 1, 0>(x, lane_partner(x), y, y);
    }
    if (r.flags & 1) w.st(w_addr<RELX>(r.dst, base), res);
}

// A run of cyclotomic squarings, one phase per squaring: the pair reduces the operand of its NEXT square from the previous products
// - the Granger-Scott update 3 xi x + 3 y + 2 z with x = R[x0] - R[x1] - R[x2], y = R[y0] - R[y1] - R[y2], z = R[z0] - R[z1] (the
// (a + b) pairs reduce a' + b' directly: the update is linear) -, keeps it when it is a coefficient of the running value, squares it.
template <class W>
BN_FN void w_fuse_sqr(W &w, const Role &r) {
    using T = typename W::T;
    T x = fe_ssub(fe_ssub(w.ld(r.src[0]), w.ld(r.src[1])), w.ld(r.src[2]));
    T y = fe_ssub(fe_ssub(w.ld(r.src[3]), w.ld(r.src[4])), w.ld(r.src[5]));
    T z = fe_ssub(w.ld(r.src[6]), w.ld(r.src[7]));
    T a = fe_lc4_par<27, 3, 3, 2>(x, lane_partner(x), y, z);
    if (r.flags & 2) w.st(r.src[8], a);
    T res = f2b_sqr_inl(a);
    if (r.flags & 1) w.st(r.dst, res);
}

// the wider combination of the Miller program: x = R[x0] + R[x1] - R[x2] - R[x3], y = R[y0] + R[y1] + R[y2] - R[y3] - R[y4]; xi x + y
template <class W>
BN_FN void w_comb2(W &w, const Role &r, uint32_t base) {
    using T = typename W::T;
    T x = fe_ssub(fe_ssub(fe_add(w.ld(w_addr<false>(r.src[0], base)), w.ld(w_addr<false>(r.src[1], base))), w.ld(w_addr<false>(r.src[2], base))), w.ld(w_addr<false>(r.src[3], base)));
    T y = fe_add(fe_add(w.ld(w_addr<false>(r.src[4], base)), w.ld(w_addr<false>(r.src[5], base))), w.ld(w_addr<false>(r.src[6], base)));
    y = fe_ssub(fe_ssub(y, w.ld(w_addr<false>(r.src[7], base))), w.ld(w_addr<false>(r.src[8], base)));
    T res = fe_lc4_par<9, 1, 1, 0>(x, lane_partner(x), y, y);
    if (r.flags & 1) w.st(w_addr<false>(r.dst, base), res);
}

template <class W>
BN_FN void w_inv(W &w, const Role &r, uint32_t base) {
    using T = typename W::T;
    Fq2B<T> a = {w.ld(w_addr<false>(r.src[0], base))};
    Fq2B<T> t = f2_inverse(a);
    if (r.flags & 1) w.st(w_addr<false>(r.dst, base), t.v);
}

// runs a program (wave_tables.hpp PROG_*): one phase per entry, a wave-level barrier after each
template <class W>
BN_FN void w_run(W &w, const uint32_t *prog) {
    // the role of the NEXT phase is fetched while this one computes: the tables live in global memory (an L2 hit of a few hundred
    // nanoseconds, a third of a phase), and the gathers that need the role are an LDS round trip of their own
    uint32_t e = prog[0];
    Role r = w.role((e >> 4) & 255u);
#pragma unroll 1
    for (int pc = 0;; ++pc) {
        const uint32_t op = e & 15u, base = e >> 12;
        if (op == OP_END) break;
        const uint32_t e_next = prog[pc + 1];
        const Role r_next = w.role((e_next >> 4) & 255u);
        if (op == OP_FUSE_SQR) w_fuse_sqr(w, r);
        else if (op == OP_PROD_SQR) w_prod<2, 0, true, false, false>(w, r, base);
        else if (op == OP_COMB_C) w_comb<true, false>(w, r, base);
        else if (op == OP_PROD_MUL) w_prod<4, 4, false, false, false>(w, r, base);
        else if (op == OP_PROD_MUL1) w_prod<1, 1, false, false, false>(w, r, base);
        else if (op == OP_PROD_MUL2) w_prod<2, 2, false, false, false>(w, r, base);
        else if (op == OP_PROD_MULR) w_prod<4, 4, false, false, true>(w, r, base);
        else if (op == OP_COMB_M) w_comb<false, false>(w, r, base);
        else if (op == OP_COMB_MR) w_comb<false, true>(w, r, base);
        else if (op == OP_COMB_M2) w_comb2(w, r, base);
        else if (op == OP_PROD_MULC) w_prod<1, 1, false, true, false>(w, r, base);
        else w_inv(w, r, base);
        w.sync();
        e = e_next; r = r_next;
    }
}

// boundary: pair j < 6 moves coefficient j of an Fq12 between its 384-byte reference image and register `reg0 + j`
template <class W>
BN_FN void w_load_f12(W &w, const uint32_t *img, uint32_t reg0_off) {
    using T = typename W::T;
    const int j = w.pair();
    if (j < 6) w.st(reg0_off + 8u * (uint32_t)j, lane_load_pair((const T *)nullptr, img + 16 * j, img + 16 * j + 8));
}
// inputs of the Miller program: P = (x, y, z) in Fq as the Fq2 values (x, 0), (y, 0), (z, 0) on pairs 0..2; Q on pairs 3..5;
// the same pairs then test z == 0 (groups/mod.rs:766: a point at infinity in either argument makes the pairing one)
template <class W>
BN_FN void w_load_points(W &w, const uint32_t *g1, const uint32_t *g2) {
    using T = typename W::T;
    const int j = w.pair();
    if (j < 3) {
        T v = lane_load_pair((const T *)nullptr, g1 + 8 * j, g1 + 8 * j);        // both lanes load the coordinate ...
        w.st((uint32_t)OFF_IN_P + 8u * (uint32_t)j, lane_pick(v, lane_bcast((const T *)nullptr, fe_zero())));     // ... the odd one keeps 0
    } else if (j < 6) {
        w.st((uint32_t)OFF_IN_Q + 8u * (uint32_t)(j - 3), lane_load_pair((const T *)nullptr, g2 + 16 * (j - 3), g2 + 16 * (j - 3) + 8));
    }
}
// RES <- one (the pairing of a point at infinity)
template <class W>
BN_FN void w_set_one(W &w) {
    using T = typename W::T;
    const int j = w.pair();
    if (j == 0) w.st((uint32_t)OFF_RES, lane_pick(lane_bcast((const T *)nullptr, fe_one()), lane_bcast((const T *)nullptr, fe_zero())));     // (1, 0)
    else if (j < 6) w.st((uint32_t)OFF_RES + 8u * (uint32_t)j, lane_bcast((const T *)nullptr, fe_zero()));
}
template <class W>
BN_FN void w_store_f12(W &w, uint32_t reg0_off, uint32_t *img) {
    const int j = w.pair();
    if (j < 6) lane_store_pair(w.ld(reg0_off + 8u * (uint32_t)j), img + 16 * j, img + 16 * j + 8);
}

}}  // namespace bn254::wv

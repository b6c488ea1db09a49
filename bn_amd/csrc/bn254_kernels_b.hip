// Lane-pair mapping (Fq2B) kernels of the BN254 pairing engine for MI355X (gfx950): TWO adjacent lanes compute one
// pairing, the even lane holding the real and the odd lane the imaginary component of every Fq2 value.  Per-lane state
// halves (an Fq12 is 54 VGPRs), so the Miller loop and the exponentiation by u run out of registers instead of private
// memory; the partner's limbs arrive by DPP (quad_perm [1,0,3,2]).  A batch of 2^16 pairings is 2048 waves = 2 per SIMD,
// which is what the integer pipe needs to be saturated (profiles/r01_ubench_valu_rates.txt).
//
// This translation unit is compiled with everything on the hot loops INLINED (fe.hpp BN_INLINE_ALL: the Fq6/Fq12-sized steps and, since
// the measurement of round 1f, the multiplier- and reduction-sized leaves): values stay in VGPRs across them, there is no argument
// marshalling (v_mov was ~45 % of the non-leaf instructions) and no caller-saved/callee-saved split of the register file.  The Miller
// loop body becomes ~160 KB of straight-line code (beyond the 64 KB instruction cache), and is still 7 % faster than the call-based
// build (profiles/r01g_*).  Only the rarely executed outer steps are calls.
#define BN_INLINE_ALL 1
#define BN_WAVES 2          // resident waves per SIMD the kernels are compiled for (256 VGPRs each).  Three (168 VGPRs) was measured in round
                            // 5: the engine's own squaring stream gains 4.4 % from a third wave when nothing spills, the real kernels compiled
                            // to 168 VGPRs LOSE 17-25 % (profiles/r05_ubench_mix_occupancy.txt, r05_occupancy_ab.txt)
// Two waves share a SIMD and the hardware arbitrates OLDEST FIRST: measured with per-wave time stamps (profiles/r01j_wave_lifetimes.txt),
// wave 0 of every SIMD ran at solo speed and finished the Miller kernel after 3.0 ms while wave 1 crawled (0.3 of the solo
// rate) and needed 5.2 ms - the SIMD ran ONE wave for the last 40 % of the kernel.  A SIMD with a privileged and a gap-filling
// wave delivers 1.3x the work of one wave, so keeping both alive to the end is worth 10 % (two concurrent processes, whose
// kernels backfill each other's freed slots, showed exactly that: 7.94 vs 7.24 M pairings/s).  Keeping the two waves in
// lockstep is NOT the answer: measured 6 % slower than doing nothing, because both streams then hit the multiplier at the same
// time; strict priority with ONE role swap is, set with s_setprio at the top of the loop steps.  Policies measured and retired
// (profiles/r03_ab_fair_policy.txt, r04p_ab_fair_policy.txt; their code left with round 5): priority by step parity; opposite priorities
// swapped every 2^20 shader cycles (the final exponentiation's policy in rounds 2-4: 0.8 % slower than the hand-over at the end of round
// 4); progress counters of the waves of a CU in LDS with the wave behind privileged (slower: the bookkeeping inside the Fq6 products
// costs more than the tail it removes); plain age arbitration (7 % slower).
#include <hip/hip_runtime.h>
// ONE hand-over: the older wave of a SIMD (slot 0) runs with priority until it has done 1/(1+r) of its steps (r = 0.3: the rate at
// which the other wave advances meanwhile), then yields for good; the younger wave takes over when it has done r/(1+r) of its
// steps - the same moment if the model holds, and the intermediate levels (1, 2) make either order of arrival safe.  Both
// waves then finish together and the SIMD always runs its efficient mode: one privileged stream, one filling the gaps.
// (Hand-over point re-swept at the end of round 4: flat between 769 and 830 per mille in both kernels, worse outside.)
constexpr int BN_B_FAIR_PERMILLE = 769;
__device__ __forceinline__ void bn_fair_handover(int step, int total) {
    const uint32_t slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 1;      // HW_ID.wave_id: 0 / 1 for the two resident waves
    if (slot == 0) { if (step * 1000 < BN_B_FAIR_PERMILLE * total) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
    else { if (step * 1000 < (1000 - BN_B_FAIR_PERMILLE) * total) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2); }
}
#define BN_MILLER_HOOK(step, total) bn_fair_handover(step, total)
#define BN_EXP_HOOK(step, total) bn_fair_handover(step, total)
#include "curve.hpp"
#include "io.hpp"

using namespace bn254;

namespace {
constexpr int BLOCK = 64;
typedef Fq2B<Fe> F2;

// Miller-loop state parked in LDS: 7 field elements x 9 limbs = 63 dwords per lane, laid out [dword][lane] so that a wave's
// ds_read/ds_write_b32 touch 64 consecutive banks (conflict-free).  16 KB per wave, 8 waves per CU = 129 KB of the 160 KB.
constexpr int PARK_DWORDS = 63;
struct MillerStateLds {
    uint32_t *base;          // this lane's column of the block's LDS array
    __device__ __forceinline__ void st_fe(int slot, const Fe &v) const {
#pragma unroll
        for (int i = 0; i < 9; ++i) base[(slot * 9 + i) * BLOCK] = v.l[i];
    }
    __device__ __forceinline__ Fe ld_fe(int slot) const {
        Fe v;
#pragma unroll
        for (int i = 0; i < 9; ++i) v.l[i] = base[(slot * 9 + i) * BLOCK];
        return v;
    }
    __device__ __forceinline__ void put_r(const G2Proj<F2> &v) const { st_fe(0, v.x.v); st_fe(1, v.y.v); st_fe(2, v.z.v); }
    __device__ __forceinline__ G2Proj<F2> get_r() const { return {{ld_fe(0)}, {ld_fe(1)}, {ld_fe(2)}}; }
    __device__ __forceinline__ void put_base(const G2Aff<F2> &v) const { st_fe(3, v.x.v); st_fe(4, v.y.v); }
    __device__ __forceinline__ G2Aff<F2> get_base() const { return {{ld_fe(3)}, {ld_fe(4)}}; }
    __device__ __forceinline__ void put_p(const G1Aff<Fe> &v) const { st_fe(5, v.x); st_fe(6, v.y); }
    __device__ __forceinline__ G1Aff<Fe> get_p() const { return {ld_fe(5), ld_fe(6)}; }
};

template <bool NAF>
__device__ __forceinline__ void miller_B_body(const uint32_t *g1, const uint32_t *g2, uint32_t *f_out, uint32_t n) {
    // (the register allocation of this kernel is sensitive to its exact shape: 3 spilled VGPRs in this form, 14 when the value
    //  computation is factored into a helper)
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;                       // keep both lanes of every pair active for the DPP exchanges
    const uint32_t *w1 = g1 + 24u * pair, *w2 = g2 + 48u * pair;
    bool inf = words_all_zero(w1 + 16, 8) || words_all_zero(w2 + 32, 16);        // groups/mod.rs:766
    G1Aff<Fe> p;
    G2Aff<F2> q;
    pair_prologue<Fe>(f2_scalar_load((const F2 *)nullptr, w1), f2_scalar_load((const F2 *)nullptr, w1 + 8), f2_scalar_load((const F2 *)nullptr, w1 + 16),
                      f2_load((const F2 *)nullptr, w2), f2_load((const F2 *)nullptr, w2 + 16), f2_load((const F2 *)nullptr, w2 + 32), p, q);
    __shared__ uint32_t park[PARK_DWORDS * BLOCK];
    MillerStateLds st = {park + threadIdx.x};
    Fq12<F2> f;
    f = miller_loop_sched<NAF>(p, q, st);
    Fq12<F2> one = f12_one<F2>();
    f.c0.c0 = f2_select(inf, f.c0.c0, one.c0.c0); f.c0.c1 = f2_select(inf, f.c0.c1, one.c0.c1); f.c0.c2 = f2_select(inf, f.c0.c2, one.c0.c2);
    f.c1.c0 = f2_select(inf, f.c1.c0, one.c1.c0); f.c1.c1 = f2_select(inf, f.c1.c1, one.c1.c1); f.c1.c2 = f2_select(inf, f.c1.c2, one.c1.c2);
    if (live) f12_store(f, f_out + 96u * pair);
}

// reference schedule: the Miller VALUES equal the reference's (bn254_miller_batch_dev, prepared-mode cross checks)
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_miller_B(const uint32_t *g1, const uint32_t *g2, uint32_t *f_out, uint32_t n) {
    miller_B_body<false>(g1, g2, f_out, n);
}
// NAF schedule (pairing.hpp miller_loop_sched<true>): used wherever a final exponentiation follows
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_miller_naf_B(const uint32_t *g1, const uint32_t *g2, uint32_t *f_out, uint32_t n) {
    miller_B_body<true>(g1, g2, f_out, n);
}

// Table of the exponentiation machine (pairing.hpp ExpTableVars) in global memory.  One Fq6 half of a slot is 27 dwords per lane,
// stored as 7 groups of 4 dwords (one pad dword): group-major, then the lane - every access of a wave is ONE global_load/store_dwordx4
// over 1 KB of contiguous memory (a dword-per-instruction layout costs 27 memory instructions per half; at two waves per SIMD every
// memory instruction costs the SIMD ~40 cycles, measured on the spills of the Miller kernel).
struct ExpTableMem {
    uint4 *table;            // wave-uniform base (SGPRs)
    uint32_t lane;           // this lane's column
    uint32_t stride;         // lanes in the launch
    // 32-bit element index (the table is far below 2^32 x 16 bytes): with a per-lane slot (Gt::pow's digit) the address is ONE VGPR
    // offset on the scalar base instead of a 64-bit pointer per access
#ifdef BN_AB_ALIAS_SCRATCH
    // TIMING EXPERIMENT ONLY (wrong results): every slot of every lane aliases slot 0 of one of 8192 lanes - a 1.8 MB footprint that
    // stays in the L2 -, so the kernel runs the same instruction stream with (almost) no HBM traffic: what the table's traffic costs
    __device__ __forceinline__ uint32_t row(int, int half) const { return (uint32_t)(half * 7) * stride + (lane & 8191u); }
#else
    __device__ __forceinline__ uint32_t row(int slot, int half) const { return (uint32_t)((slot * 2 + half) * 7) * stride + lane; }
#endif
    __device__ __forceinline__ void st6(int slot, int half, const Fq6<F2> &v) const {
        const uint32_t r = row(slot, half);
        uint32_t w[28];
#pragma unroll
        for (int i = 0; i < 9; ++i) { w[i] = v.c0.v.l[i]; w[9 + i] = v.c1.v.l[i]; w[18 + i] = v.c2.v.l[i]; }
        w[27] = 0;
#pragma unroll
        for (int g = 0; g < 7; ++g) table[r + (uint32_t)g * stride] = make_uint4(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]);
    }
    __device__ __forceinline__ Fq6<F2> ld6(int slot, int half) const {
        const uint32_t r = row(slot, half);
        uint32_t w[28];
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            const uint4 x = table[r + (uint32_t)g * stride];
            w[4 * g] = x.x; w[4 * g + 1] = x.y; w[4 * g + 2] = x.z; w[4 * g + 3] = x.w;
        }
        Fq6<F2> v;
#pragma unroll
        for (int i = 0; i < 9; ++i) { v.c0.v.l[i] = w[i]; v.c1.v.l[i] = w[9 + i]; v.c2.v.l[i] = w[18 + i]; }
        return v;
    }
    __device__ __forceinline__ void put(int i, const Fq12<F2> &v) const { st6(i, 0, v.c0); st6(i, 1, v.c1); }
    __device__ __forceinline__ Fq6<F2> c0(int i) const { return ld6(i, 0); }
    __device__ __forceinline__ Fq6<F2> c1(int i) const { return ld6(i, 1); }
};
constexpr size_t EXP_TABLE_DWORDS_PER_LANE = (size_t)k::EXP_SLOTS * 2 * 7 * 4;

__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_final_exp_B(const uint32_t *f_in, uint32_t *out, uint32_t n, uint32_t *table) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    ExpTableMem tbl = {(uint4 *)table, t, gridDim.x * BLOCK};
    Fq12<F2> f = final_exponentiation(f12_load<F2>(f_in + 96u * pair), tbl);
    if (live) f12_store(f, out + 96u * pair);
}
// ---- the multi-pairing's Miller loop with a shared accumulator (pairing.hpp miller_loop_shared): M pairs per lane pair
// State of the M pairs in global memory: per lane and pair 17 x 16 bytes - R (27 dwords in 7 groups of 4), the base point (18 in 5),
// P (18 in 5) -, group-major, then the lane: every access of a wave is one coalesced dwordx4 instruction over 1 KB (ExpTableMem's
// layout).  Both lanes of a pair hold P (an Fq point) replicated.
// The M affine P_i live in LDS ([pair][dword][lane]: 72 B per lane and pair, 18 KB per workgroup at M = 4, eight workgroups per CU = 147 of
// the 160 KB): they are read in EVERY step of every pair (in global memory: 2.2 % slower, profiles/r04e_ab_shared_p_lds.txt)
template <int M>
struct MillerStateMem {
    uint4 *base;             // wave-uniform
    uint32_t lane, stride;   // this lane's column, lanes in the launch
    uint32_t infmask;        // bit i: pair i is (treated as) infinite
    uint32_t *plds;          // this lane's column of the workgroup's P store
#ifdef BN_AB_ALIAS_SCRATCH
    __device__ __forceinline__ uint32_t row(int, int g) const { return (uint32_t)g * stride + (lane & 8191u); }      // timing experiment, see ExpTableMem
#else
    __device__ __forceinline__ uint32_t row(int i, int g) const { return (uint32_t)(i * 17 + g) * stride + lane; }
#endif
    template <int N, int G0>
    __device__ __forceinline__ void st_n(int i, const Fe *v) const {             // N field elements -> ceil(9N/4) groups from group G0
        uint32_t w[((9 * N + 3) / 4) * 4];
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int l = 0; l < 9; ++l) w[9 * k + l] = v[k].l[l];
#pragma unroll
        for (int l = 9 * N; l < ((9 * N + 3) / 4) * 4; ++l) w[l] = 0;
#pragma unroll
        for (int g = 0; g < (9 * N + 3) / 4; ++g) base[row(i, G0 + g)] = make_uint4(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]);
    }
    template <int N, int G0>
    __device__ __forceinline__ void ld_n(int i, Fe *v) const {
        uint32_t w[((9 * N + 3) / 4) * 4];
#pragma unroll
        for (int g = 0; g < (9 * N + 3) / 4; ++g) {
            const uint4 x = base[row(i, G0 + g)];
            w[4 * g] = x.x; w[4 * g + 1] = x.y; w[4 * g + 2] = x.z; w[4 * g + 3] = x.w;
        }
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int l = 0; l < 9; ++l) v[k].l[l] = w[9 * k + l];
    }
    __device__ __forceinline__ void put_r(int i, const G2Proj<F2> &v) const { Fe t[3] = {v.x.v, v.y.v, v.z.v}; st_n<3, 0>(i, t); }
    __device__ __forceinline__ G2Proj<F2> get_r(int i) const { Fe t[3]; ld_n<3, 0>(i, t); return {{t[0]}, {t[1]}, {t[2]}}; }
    __device__ __forceinline__ void put_base(int i, const G2Aff<F2> &v) const { Fe t[2] = {v.x.v, v.y.v}; st_n<2, 7>(i, t); }
    __device__ __forceinline__ G2Aff<F2> get_base(int i) const { Fe t[2]; ld_n<2, 7>(i, t); return {{t[0]}, {t[1]}}; }
    __device__ __forceinline__ void put_p(int i, const G1Aff<Fe> &v) const {
#pragma unroll
        for (int l = 0; l < 9; ++l) { plds[((i * 18) + l) * BLOCK] = v.x.l[l]; plds[((i * 18) + 9 + l) * BLOCK] = v.y.l[l]; }
    }
    __device__ __forceinline__ G1Aff<Fe> get_p(int i) const {
        G1Aff<Fe> v;
#pragma unroll
        for (int l = 0; l < 9; ++l) { v.x.l[l] = plds[((i * 18) + l) * BLOCK]; v.y.l[l] = plds[((i * 18) + 9 + l) * BLOCK]; }
        return v;
    }
    __device__ __forceinline__ bool is_inf(int i) const { return (infmask >> i) & 1u; }
};
constexpr size_t MILLER_STATE_BYTES_PER_LANE_AND_PAIR = 17 * 16;

// f_out[t] = prod_{i < M} miller(p[M t + i], q[M t + i])  (pairs beyond n count as infinity -> 1); the values only meet a final exponentiation
template <int M>
__device__ __forceinline__ void miller_shared_body(const uint32_t *g1, const uint32_t *g2, uint32_t *f_out, uint32_t n, uint4 *state) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t lp = t >> 1, groups = (n + M - 1) / M;
    const bool live = lp < groups;
    __shared__ uint32_t p_store[M * 18 * BLOCK];
    MillerStateMem<M> st = {state, t, gridDim.x * BLOCK, 0u, p_store + threadIdx.x};
#pragma unroll 1
    for (int i = 0; i < M; ++i) {
        uint32_t pair = (live ? lp : groups - 1) * M + (uint32_t)i;
        const bool beyond = pair >= n;
        if (beyond) pair = n - 1;                                   // keep both lanes active for the DPP exchanges
        const uint32_t *w1 = g1 + 24u * pair, *w2 = g2 + 48u * pair;
        if (beyond || words_all_zero(w1 + 16, 8) || words_all_zero(w2 + 32, 16)) st.infmask |= 1u << i;        // groups/mod.rs:766
        G1Aff<Fe> p;
        G2Aff<F2> q;
        pair_prologue<Fe>(f2_scalar_load((const F2 *)nullptr, w1), f2_scalar_load((const F2 *)nullptr, w1 + 8), f2_scalar_load((const F2 *)nullptr, w1 + 16),
                          f2_load((const F2 *)nullptr, w2), f2_load((const F2 *)nullptr, w2 + 16), f2_load((const F2 *)nullptr, w2 + 32), p, q);
        const Fe t2 = f2_scalar_const((const F2 *)nullptr, k::ISO_T2), t3 = f2_scalar_const((const F2 *)nullptr, k::ISO_T3);      // onto the isomorphic curve
        p = {fe_mul(p.x, t2), fe_mul(p.y, t3)};
        q = {f2_scale(q.x, t2), f2_scale(q.y, t3)};
        st.put_p(i, p); st.put_base(i, q);
        st.put_r(i, G2Proj<F2>{q.x, q.y, f2_one((const F2 *)nullptr)});
    }
    Fq12<F2> f = miller_loop_shared<M, F2, Fe>(st);
    if (live) f12_store(f, f_out + 96u * lp);
}
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_miller_shared2_B(const uint32_t *g1, const uint32_t *g2, uint32_t *f_out, uint32_t n, uint4 *state) {
    miller_shared_body<2>(g1, g2, f_out, n, state);
}
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_miller_shared4_B(const uint32_t *g1, const uint32_t *g2, uint32_t *f_out, uint32_t n, uint4 *state) {
    miller_shared_body<4>(g1, g2, f_out, n, state);
}

// ---- prepared-G2 mode: 102 line coefficients per Q, 48 u32 each (ell_0, ell_vw, ell_vv as Fq2 in the reference image)
constexpr int NCOEFF = 102, COEFF_WORDS = 48;
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_g2_precompute_B(const uint32_t *g2, uint32_t *coeffs, uint32_t n) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    const uint32_t *w2 = g2 + 48u * pair;
    G2Aff<F2> q = g2_to_affine(f2_load((const F2 *)nullptr, w2), f2_load((const F2 *)nullptr, w2 + 16), f2_load((const F2 *)nullptr, w2 + 32));
    uint32_t *dst = coeffs + (size_t)pair * NCOEFF * COEFF_WORDS;
    auto sink = [&](int idx, const Line<F2> &l) {
        if (live) {
            uint32_t *c = dst + idx * COEFF_WORDS;
            f2_store(l.ell_0, c); f2_store(l.ell_vw, c + 16); f2_store(l.ell_vv, c + 32);
        }
    };
    precompute_lines(q, sink);
}
// f[i] = miller_loop(coeffs, P[i])  (groups/mod.rs:486-519); coeff_stride = 0 shares one coefficient set among all P
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_miller_prepared_B(const uint32_t *g1, const uint32_t *coeffs, uint32_t coeff_stride, uint32_t *f_out, uint32_t n) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    const uint32_t *w1 = g1 + 24u * pair;
    bool inf = words_all_zero(w1 + 16, 8);
    Fe zi = fe_inverse(f2_scalar_load((const F2 *)nullptr, w1 + 16)), zi2 = fe_sqr(zi);
    G1Aff<Fe> p = {fe_mul(f2_scalar_load((const F2 *)nullptr, w1), zi2), fe_mul(f2_scalar_load((const F2 *)nullptr, w1 + 8), fe_mul(zi2, zi))};
    const uint32_t *src = coeffs + (size_t)pair * coeff_stride;
    auto source = [&](int idx) {
        const uint32_t *c = src + idx * COEFF_WORDS;
        Line<F2> l = {f2_load((const F2 *)nullptr, c), f2_load((const F2 *)nullptr, c + 16), f2_load((const F2 *)nullptr, c + 32)};
        return l;
    };
    __shared__ uint32_t park[18 * BLOCK];
    MillerStateLds st = {park + threadIdx.x};            // only slots 0, 1 are used here: P.x, P.y
    st.st_fe(0, p.x); st.st_fe(1, p.y);
    struct PInLds { const MillerStateLds &s; __device__ __forceinline__ G1Aff<Fe> get_p() const { return {s.ld_fe(0), s.ld_fe(1)}; } } ps = {st};
    Fq12<F2> f = miller_loop_prepared<F2>(ps, source);
    Fq12<F2> one = f12_one<F2>();
    f.c0.c0 = f2_select(inf, f.c0.c0, one.c0.c0); f.c0.c1 = f2_select(inf, f.c0.c1, one.c0.c1); f.c0.c2 = f2_select(inf, f.c0.c2, one.c0.c2);
    f.c1.c0 = f2_select(inf, f.c1.c0, one.c1.c0); f.c1.c1 = f2_select(inf, f.c1.c1, one.c1.c1); f.c1.c2 = f2_select(inf, f.c1.c2, one.c1.c2);
    if (live) f12_store(f, f_out + 96u * pair);
}

// ---- NATIVE prepared-G2 mode (pairing.hpp precompute_native / miller_loop_native): NATIVE_LINES records per Q, one record = 12 x 16 bytes per
// lane of the pair: groups 0-6 hold A (this lane's component), xi B u, xi B v (27 dwords + 1 pad), groups 7-11 B u, B v (18 + 2 pad), all in the
// 9 x 29-bit radix-2^261 limbs the multiplier takes.  Group-major, then the column (2 x Q index + lane parity): with one table per pairing a
// wave's access is one coalesced dwordx4 instruction over 1 KB; with ONE shared Q every lane of a parity reads the same 16 bytes (stride 2
// columns: a broadcast out of the L1 / L2, 33.8 KB per Q for the whole launch).  33 792 B per Q.
constexpr int NATIVE_GROUPS = 12;
struct NativeTableMem {
    uint4 *base;             // wave-uniform
    uint32_t col, stride;    // this lane's column, columns in the table (2 x number of Q)
    __device__ __forceinline__ uint4 *row(int line, int g) const { return base + (size_t)(uint32_t)(line * NATIVE_GROUPS + g) * stride; }
    template <int N, int G0>
    __device__ __forceinline__ void st_n(int line, const Fe *v) const {
        uint32_t w[((9 * N + 3) / 4) * 4];
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int l = 0; l < 9; ++l) w[9 * k + l] = v[k].l[l];
#pragma unroll
        for (int l = 9 * N; l < ((9 * N + 3) / 4) * 4; ++l) w[l] = 0;
#pragma unroll
        for (int g = 0; g < (9 * N + 3) / 4; ++g) row(line, G0 + g)[col] = make_uint4(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]);
    }
    template <int N, int G0>
    __device__ __forceinline__ void ld_n(int line, Fe *v) const {
        uint32_t w[((9 * N + 3) / 4) * 4];
#pragma unroll
        for (int g = 0; g < (9 * N + 3) / 4; ++g) {
            const uint4 x = row(line, G0 + g)[col];
            w[4 * g] = x.x; w[4 * g + 1] = x.y; w[4 * g + 2] = x.z; w[4 * g + 3] = x.w;
        }
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int l = 0; l < 9; ++l) v[k].l[l] = w[9 * k + l];
    }
    // between the two passes of precompute_native a record parks ell_0, d, c (groups 0-6) and the running product of the d (groups 7-9)
    __device__ __forceinline__ void put_raw(int i, const F2 &e0, const F2 &d, const F2 &c, const F2 &pref) const {
        Fe t[3] = {e0.v, d.v, c.v}; st_n<3, 0>(i, t);
        st_n<1, 7>(i, &pref.v);
    }
    __device__ __forceinline__ void get_raw(int i, F2 &e0, F2 &d, F2 &c) const { Fe t[3]; ld_n<3, 0>(i, t); e0.v = t[0]; d.v = t[1]; c.v = t[2]; }
    __device__ __forceinline__ F2 get_prefix(int i) const { Fe t; ld_n<1, 7>(i, &t); return {t}; }
    __device__ __forceinline__ void put_final(int i, const Fe &a, const Fq2BPrep<Fe> &b, const Fq2BPrep<Fe> &xb) const {
        Fe t[3] = {a, xb.u, xb.v}; st_n<3, 0>(i, t);
        Fe u[2] = {b.u, b.v}; st_n<2, 7>(i, u);
    }
};
// table[.., 2 pair + lane] <- the native lines of q[pair]; q_inf[pair] <- (q[pair] is infinity: the pairing is then one, groups/mod.rs:766).
// Not a hot path: one launch per set of Q, ~0.5 Miller loops of work per Q (two passes around ONE inversion).  Lanes beyond n repeat the last
// pair and store the same bytes to the same addresses (every lane re-reads only what it wrote itself).
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_g2_prepare_native_B(const uint32_t *g2, uint4 *table, uint32_t stride, uint32_t *q_inf, uint32_t n) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    if (pair >= n) pair = n - 1;
    const uint32_t *w2 = g2 + 48u * pair;
    if ((threadIdx.x & 1u) == 0) q_inf[pair] = words_all_zero(w2 + 32, 16) ? 1u : 0u;
    G2Aff<F2> q = g2_to_affine(f2_load((const F2 *)nullptr, w2), f2_load((const F2 *)nullptr, w2 + 16), f2_load((const F2 *)nullptr, w2 + 32));
    NativeTableMem st = {table, 2u * pair + (threadIdx.x & 1u), stride};
    precompute_native(q, st);
}
// f[i] = Miller value of (p[i], Q) over the native table (pairing.hpp miller_loop_native): only ever meets a final exponentiation.
// shared != 0: every pairing reads column pair 0; else pairing i reads the table of Q number q_lo + i.
// Per pairing in LDS ([dword][lane], like the fused loop's parked state): sigma, tau, 9 tau, -+tau = 36 dwords per lane, 9 KB per wave.
struct NativeLineMem {
    const uint4 *base;
    uint32_t col, stride;
    uint32_t *lds;           // this lane's column of the block's LDS array
    int line;
    mutable Fe xbu, xbv;     // fetched together with A (same 16-byte groups)
    __device__ __forceinline__ Fe ld_lds(int slot) const {
        Fe v;
#pragma unroll
        for (int i = 0; i < 9; ++i) v.l[i] = lds[(slot * 9 + i) * BLOCK];
        return v;
    }
    __device__ __forceinline__ void st_lds(int slot, const Fe &v) const {
#pragma unroll
        for (int i = 0; i < 9; ++i) lds[(slot * 9 + i) * BLOCK] = v.l[i];
    }
    __device__ __forceinline__ void set_line(int i) { line = i; }
    __device__ __forceinline__ Fq2BPrep<Fe> x0() const {
        const NativeTableMem t = {const_cast<uint4 *>(base), col, stride};
        Fe v[3];
        t.ld_n<3, 0>(line, v);
        xbu = v[1]; xbv = v[2];
        return f2b_prepare(F2{fe_mul(v[0], ld_lds(0))});
    }
    __device__ __forceinline__ Fq2BPrep<Fe> xb() const { return {xbu, xbv}; }
    __device__ __forceinline__ Fq2BPrep<Fe> b() const {
        const NativeTableMem t = {const_cast<uint4 *>(base), col, stride};
        Fe v[2];
        t.ld_n<2, 7>(line, v);
        return {v[0], v[1]};
    }
    __device__ __forceinline__ Fe tau() const { return ld_lds(1); }
    __device__ __forceinline__ Fe tau9() const { return ld_lds(2); }
    __device__ __forceinline__ Fe taum() const { return ld_lds(3); }
};
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_miller_native_B(const uint32_t *g1, const uint4 *table, uint32_t stride, uint32_t q_lo, int shared, const uint32_t *q_inf, uint32_t *f_out, uint32_t n) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    const uint32_t *w1 = g1 + 24u * pair;
    const uint32_t qi = shared ? 0u : q_lo + pair;
    const bool inf = words_all_zero(w1 + 16, 8) || q_inf[qi] != 0u;                 // groups/mod.rs:766
    __shared__ uint32_t park[36 * BLOCK];
    NativeLineMem src = {table, 2u * qi + (threadIdx.x & 1u), stride, park + threadIdx.x, 0, {}, {}};
    {
        const PNative<Fe> pn = p_native(f2_scalar_load((const F2 *)nullptr, w1), f2_scalar_load((const F2 *)nullptr, w1 + 8), f2_scalar_load((const F2 *)nullptr, w1 + 16));
        src.st_lds(0, pn.sigma); src.st_lds(1, pn.tau); src.st_lds(2, pn.tau9); src.st_lds(3, pn.taum);
    }
    Fq12<F2> f = miller_loop_native<F2>(src);
    Fq12<F2> one = f12_one<F2>();
    f.c0.c0 = f2_select(inf, f.c0.c0, one.c0.c0); f.c0.c1 = f2_select(inf, f.c0.c1, one.c0.c1); f.c0.c2 = f2_select(inf, f.c0.c2, one.c0.c2);
    f.c1.c0 = f2_select(inf, f.c1.c0, one.c1.c0); f.c1.c1 = f2_select(inf, f.c1.c1, one.c1.c1); f.c1.c2 = f2_select(inf, f.c1.c2, one.c1.c2);
    if (live) f12_store(f, f_out + 96u * pair);
}

// Every native table ends with the IDENTITY column pair (column 2 nq, 2 nq + 1): A = 1, B = xi B = 0 in all 88 lines, so that a lane pair reading
// it with sigma = 1, tau = 0 multiplies its accumulator by exactly one - how the shared-accumulator loop below treats a point at infinity.
__global__ void __launch_bounds__(BLOCK) bn254_native_identity_B(uint4 *table, uint32_t stride) {
    for (uint32_t r = threadIdx.x; r < (uint32_t)(NATIVE_LINES * NATIVE_GROUPS); r += BLOCK) {
        const uint32_t g = r % NATIVE_GROUPS;
        uint4 even = make_uint4(0, 0, 0, 0);                                       // A own = ONE on the even lane (9 limbs: groups 0, 1, 2), zero elsewhere
        if (g == 0) even = make_uint4(k::ONE[0], k::ONE[1], k::ONE[2], k::ONE[3]);
        if (g == 1) even = make_uint4(k::ONE[4], k::ONE[5], k::ONE[6], k::ONE[7]);
        if (g == 2) even = make_uint4(k::ONE[8], 0, 0, 0);
        table[(size_t)r * stride + stride - 2] = even;
        table[(size_t)r * stride + stride - 1] = make_uint4(0, 0, 0, 0);
    }
}
// ---- the multi-pairing over native tables (pairing.hpp miller_loop_native_shared): M pairs per lane pair on ONE accumulator, pair i of lane pair
// t is pairing t + i ceil(n / M) of the launch and reads the table of point q_lo + t + i ceil(n / M) (shared: of point 0).  Per pair in LDS: sigma, tau = 18 dwords
// per lane ([pair][dword][lane]; 18 KB per workgroup at M = 4, eight workgroups per CU = 147 of the 160 KB).  There is no per-step point state.
template <int M>
struct NativeSharedMem {
    const uint4 *base;
    uint32_t col0, ident, stride, infmask;   // column of pair 0, the identity column (both incl. the lane's parity; shared: col0 for every pair)
    uint32_t *lds;
    int line, cur, col_step;
    mutable Fe xbu, xbv;
    __device__ __forceinline__ uint32_t col() const { return ((infmask >> cur) & 1u) ? ident : col0 + (uint32_t)(cur * col_step); }
    __device__ __forceinline__ Fe ld_lds(int slot) const {
        Fe v;
#pragma unroll
        for (int i = 0; i < 9; ++i) v.l[i] = lds[((cur * 2 + slot) * 9 + i) * BLOCK];
        return v;
    }
    __device__ __forceinline__ void st_lds(int pair, int slot, const Fe &v) const {
#pragma unroll
        for (int i = 0; i < 9; ++i) lds[((pair * 2 + slot) * 9 + i) * BLOCK] = v.l[i];
    }
    __device__ __forceinline__ void set_line(int l, int i) { line = l; cur = i; }
    __device__ __forceinline__ Fq2BPrep<Fe> x0() const {
        Fe v[3];
        const NativeTableMem t = {const_cast<uint4 *>(base), col(), stride};
        t.ld_n<3, 0>(line, v);
        xbu = v[1]; xbv = v[2];
        return f2b_prepare(F2{fe_mul(v[0], ld_lds(0))});
    }
    __device__ __forceinline__ Fq2BPrep<Fe> xb() const { return {xbu, xbv}; }
    __device__ __forceinline__ Fq2BPrep<Fe> b() const {
        const NativeTableMem t = {const_cast<uint4 *>(base), col(), stride};
        Fe v[2];
        t.ld_n<2, 7>(line, v);
        return {v[0], v[1]};
    }
    __device__ __forceinline__ Fe tau() const { return ld_lds(1); }
    __device__ __forceinline__ Fe tau9() const { return p_native_tau9(ld_lds(1)); }
    __device__ __forceinline__ Fe taum() const { return p_native_taum(ld_lds(1)); }
};
template <int M>
__device__ __forceinline__ void miller_native_shared_body(const uint32_t *g1, const uint4 *table, uint32_t stride, uint32_t q_lo, int shared, const uint32_t *q_inf, uint32_t *f_out, uint32_t n) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t lp = t >> 1, groups = (n + M - 1) / M;
    const bool live = lp < groups;
    // pair i of lane pair t is pairing t + i * groups of the launch: for a fixed i the lane pairs of a wave read ADJACENT table columns (one coalesced
    // 1 KB row per load instruction; with pairs M t + i a row would be fetched M times, 22 us apart - measured 18 % slower on 2^18 tables)
    const uint32_t lp0 = live ? lp : groups - 1, parity = threadIdx.x & 1u;
    __shared__ uint32_t park[M * 2 * 9 * BLOCK];
    NativeSharedMem<M> src = {table, shared ? parity : 2u * (q_lo + lp0) + parity, stride - 2u + parity, stride, 0u, park + threadIdx.x, 0, 0, shared ? 0 : (int)(2u * groups), {}, {}};
#pragma unroll 1
    for (int i = 0; i < M; ++i) {
        uint32_t pair = lp0 + (uint32_t)i * groups;
        const bool beyond = pair >= n;
        if (beyond) pair = n - 1;
        const uint32_t *w1 = g1 + 24u * pair;
        const bool inf = beyond || words_all_zero(w1 + 16, 8) || q_inf[shared ? 0u : q_lo + pair] != 0u;                 // groups/mod.rs:766
        if (inf) src.infmask |= 1u << i;
        const PNative<Fe> pn = p_native(f2_scalar_load((const F2 *)nullptr, w1), f2_scalar_load((const F2 *)nullptr, w1 + 8), f2_scalar_load((const F2 *)nullptr, w1 + 16));
        Fe one, zero;
        p_native_identity(one, zero);
        src.st_lds(i, 0, fe_select(inf, pn.sigma, one)); src.st_lds(i, 1, fe_select(inf, pn.tau, zero));
    }
    Fq12<F2> f = miller_loop_native_shared<M, F2>(src);
    if (live) f12_store(f, f_out + 96u * lp);
}
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_miller_native_shared2_B(const uint32_t *g1, const uint4 *table, uint32_t stride, uint32_t q_lo, int shared, const uint32_t *q_inf, uint32_t *f_out, uint32_t n) {
    miller_native_shared_body<2>(g1, table, stride, q_lo, shared, q_inf, f_out, n);
}
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_miller_native_shared4_B(const uint32_t *g1, const uint4 *table, uint32_t stride, uint32_t q_lo, int shared, const uint32_t *q_inf, uint32_t *f_out, uint32_t n) {
    miller_native_shared_body<4>(g1, table, stride, q_lo, shared, q_inf, f_out, n);
}

// out[i] = a[i] * b[i]   (Gt * Gt, lib.rs:175-179 -> fq12.rs:295-307)
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_gt_mul_B(const uint32_t *a, const uint32_t *b, uint32_t *out, uint32_t n) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    Fq12<F2> r = f12_mul_o(f12_load<F2>(a + 96u * pair), f12_load<F2>(b + 96u * pair));
    if (live) f12_store(r, out + 96u * pair);
}
// out[i] = a[i] ^ k[i]   (Gt::pow, lib.rs:171 -> fields/mod.rs:35-46: 256 x { res = res^2; if bit { res = a * res } } on the scalar
// taken out of Montgomery form).  The power is a unique field element, so any addition chain returns the reference's bytes:
// fixed 4-bit windows, MSB first - 256 squarings + 64 table products + a 14-operation table instead of 256 x (square, multiply,
// select).  General Fq12 squarings (not cyclotomic ones): correct for ANY non-zero Fq12, like the reference's generic pow.
// The window table lives in global memory in the layout of the exponentiation machine's table (ExpTableMem: entry-major, 16 bytes
// per lane and access): the table is BUILT with wave-uniform entry numbers (perfectly coalesced dwordx4 stores) and READ with each
// lane pair's own digit - at most 9 (16) different 1 KB rows per instruction instead of 64 different cache lines with the round-2
// [lane][entry] layout.  Inputs in the cyclotomic subgroup (checked on the device: every value the reference's API can produce)
// take the signed-window Granger-Scott chain (pairing.hpp gt_pow_cyclotomic), anything else the general one.
// [lane][entry][half][7 x 16 bytes]: the entry a lane reads depends on ITS digit, so contiguous 224-byte runs per lane (two cache
// lines each) beat the entry-major rows (every lane's 16 bytes from a different 1 KB row: 4x the traffic, profiles/r03m_pmc_side.txt)
struct PowTableLane {
    uint4 *base;             // this lane's 33 x 2 x 7 vectors
#ifdef BN_AB_ALIAS_SCRATCH
    // TIMING EXPERIMENT ONLY (wrong results): the 33 entries of a lane alias entries 0 / 1 of the SAME lane - no lane writes another
    // lane's lines (aliasing whole tables made thousands of waves fight over the same lines: 3.8 -> 6.9 ms), the footprint is 1/16
    static __device__ __forceinline__ int alias(int slot) { return slot & 1; }
#else
    static __device__ __forceinline__ int alias(int slot) { return slot; }
#endif
    __device__ __forceinline__ void st6(int slot_, int half, const Fq6<F2> &v) const {
        const int slot = alias(slot_);
        uint4 *p = base + (uint32_t)(slot * 2 + half) * 7u;
        uint32_t w[28];
#pragma unroll
        for (int i = 0; i < 9; ++i) { w[i] = v.c0.v.l[i]; w[9 + i] = v.c1.v.l[i]; w[18 + i] = v.c2.v.l[i]; }
        w[27] = 0;
#pragma unroll
        for (int g = 0; g < 7; ++g) p[g] = make_uint4(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]);
    }
    __device__ __forceinline__ Fq6<F2> ld6(int slot_, int half) const {
        const int slot = alias(slot_);
        const uint4 *p = base + (uint32_t)(slot * 2 + half) * 7u;
        uint32_t w[28];
#pragma unroll
        for (int g = 0; g < 7; ++g) { const uint4 x = p[g]; w[4 * g] = x.x; w[4 * g + 1] = x.y; w[4 * g + 2] = x.z; w[4 * g + 3] = x.w; }
        Fq6<F2> v;
#pragma unroll
        for (int i = 0; i < 9; ++i) { v.c0.v.l[i] = w[i]; v.c1.v.l[i] = w[9 + i]; v.c2.v.l[i] = w[18 + i]; }
        return v;
    }
    __device__ __forceinline__ void put(int i, const Fq12<F2> &v) const { st6(i, 0, v.c0); st6(i, 1, v.c1); }
    __device__ __forceinline__ Fq6<F2> c0(int i) const { return ld6(i, 0); }
    __device__ __forceinline__ Fq6<F2> c1(int i) const { return ld6(i, 1); }
};
constexpr size_t POW_TABLE_DWORDS_PER_LANE = GT_GLS_ENTRIES * 2 * 7 * 4;          // 33 entries of 216 B: 7.4 KB per lane
// the general chain as a real function: it is the rare path, and inlined next to the cyclotomic chain the two were register-allocated
// together (97 spilled VGPRs)
typedef PowTableLane PowTable;
__device__ __noinline__ void gt_pow_general_cold(const Fq12<F2> *base, const uint32_t *raw, const PowTable *tbl, Fq12<F2> *res) {
    PowTable t = *tbl;
    *res = gt_pow_windowed(*base, raw, t);
}
__device__ __noinline__ void gt_pow_strict_cold(const Fq12<F2> *base, const uint32_t *raw, const PowTable *tbl, Fq12<F2> *res) {
    PowTable t = *tbl;
    *res = gt_pow_cyclotomic(*base, raw, t);
}
__device__ __noinline__ void gt_pow_gls_table_cold(const Fq12<F2> *base, const PowTable *tbl) {
    PowTable t = *tbl;
    gt_pow_gls_table(*base, t);
}
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_gt_pow_B(const uint32_t *a, const uint32_t *k, uint32_t *out, uint32_t n, uint32_t *table, int mode) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    // the scalar is fetched and taken out of Montgomery form where it is first needed - AFTER the table construction on the main path:
    // live across it, its eight words (and the address they came from) were spilled
    auto load_scalar = [&](uint32_t *raw) {
        uint32_t kw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) kw[i] = k[8u * pair + i];
        fr_from_mont(kw, raw);
    };
    uint32_t raw[8];
    PowTable tbl = {(uint4 *)table + (size_t)t * (POW_TABLE_DWORDS_PER_LANE / 4)};
    const Fq12<F2> base = f12_load<F2>(a + 96u * pair);
    Fq12<F2> res;
    // wave-uniform choice.  mode 0: Frobenius decomposition when every element of the wave is cyclotomic (exact for values of order
    // r - all the reference's Gt can hold), 2: the one-dimensional signed-window chain (exact for ANY cyclotomic element), 1 or a
    // non-cyclotomic element in the wave: the general chain (exact for any Fq12)
    const bool cyc = mode != 1 && __all(gt_is_cyclotomic(base));
    if (cyc && mode == 0) { gt_pow_gls_table_cold(&base, &tbl); load_scalar(raw); res = gt_pow_gls_loop<F2>(raw, tbl); }
    else if (cyc) { load_scalar(raw); gt_pow_strict_cold(&base, raw, &tbl, &res); }
    else { load_scalar(raw); gt_pow_general_cold(&base, raw, &tbl, &res); }
    if (live) f12_store(res, out + 96u * pair);
}
// out[i] = a[i]^-1   (Gt::inverse, lib.rs:172 -> fq12.rs:284-292; one Fq inversion per element, constant-time divsteps)
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_gt_inverse_B(const uint32_t *a, uint32_t *out, uint32_t n) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    Fq12<F2> r = f12_inverse(f12_load<F2>(a + 96u * pair));
    if (live) f12_store(r, out + 96u * pair);
}
// out[i] = a[i].exp_by_neg_z() exactly as the reference writes it (fields/fq12.rs:229-246: 62 x { cyclotomic_squared; multiply on the
// set bits of u }, then unitary_inverse) for ANY Fq12 - off the cyclotomic subgroup the Granger-Scott "squaring" is not a square, so the
// result is specific to this operation sequence.  The engine's own exponentiation (fe_step programs, signed digits) equals it only on
// cyclotomic elements, which is all a pairing ever feeds it; this kernel exists so that the reference's known answer for the function
// (fields/mod.rs:171-201, an element OFF the subgroup) runs on the device literally.
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_exp_by_neg_z_B(const uint32_t *a, uint32_t *out, uint32_t n) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    Fq12<F2> r = exp_by_neg_z_reference_schedule(f12_load<F2>(a + 96u * pair));
    if (live) f12_store(r, out + 96u * pair);
}
}  // namespace

extern "C" {
int bn254_launch_exp_by_neg_z_B(const void *a, void *out, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_exp_by_neg_z_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)a, (uint32_t *)out, (uint32_t)n);
    return (int)hipGetLastError();
}
int bn254_launch_g2_precompute_B(const void *q, void *coeffs, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_g2_precompute_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)q, (uint32_t *)coeffs, (uint32_t)n);
    return (int)hipGetLastError();
}
int bn254_launch_miller_prepared_B(const void *p, const void *coeffs, int shared, void *f, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_miller_prepared_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint32_t *)coeffs,
                       (uint32_t)(shared ? 0 : NCOEFF * COEFF_WORDS), (uint32_t *)f, (uint32_t)n);
    return (int)hipGetLastError();
}
// nq points + the identity column pair the table ends with
size_t bn254_native_table_bytes_B(size_t nq) { return (nq + 1) * (size_t)NATIVE_LINES * NATIVE_GROUPS * 2 * sizeof(uint4); }
int bn254_native_lines_B(void) { return NATIVE_LINES; }
// table: bn254_native_table_bytes_B(nq) bytes for nq points, q_inf: nq words
int bn254_launch_g2_prepare_native_B(const void *q, void *table, void *q_inf, size_t nq, hipStream_t s) {
    unsigned grid = (unsigned)((2 * nq + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_g2_prepare_native_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)q, (uint4 *)table, (uint32_t)(2 * nq + 2), (uint32_t *)q_inf, (uint32_t)nq);
    hipLaunchKernelGGL(bn254_native_identity_B, dim3(1), dim3(BLOCK), 0, s, (uint4 *)table, (uint32_t)(2 * nq + 2));
    return (int)hipGetLastError();
}
// nq: points in the table (its stride); shared: every p[i] against point 0, else p[i] against point q_lo + i
int bn254_launch_miller_native_B(const void *p, const void *table, const void *q_inf, size_t nq, size_t q_lo, int shared, void *f, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_miller_native_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint4 *)table, (uint32_t)(2 * nq + 2), (uint32_t)q_lo, shared ? 1 : 0,
                       (const uint32_t *)q_inf, (uint32_t *)f, (uint32_t)n);
    return (int)hipGetLastError();
}
// m pairs per lane pair on one accumulator (m = 2 or 4): G = ceil(n / m) Miller values out, value t = prod_{i < m} miller(p[t + i G], point q_lo + t + i G)
int bn254_launch_miller_native_shared_B(const void *p, const void *table, const void *q_inf, size_t nq, size_t q_lo, int shared, void *f, size_t n, int m, hipStream_t s) {
    const size_t groups = (n + m - 1) / m;
    unsigned grid = (unsigned)((2 * groups + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(m == 4 ? bn254_miller_native_shared4_B : bn254_miller_native_shared2_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint4 *)table,
                       (uint32_t)(2 * nq + 2), (uint32_t)q_lo, shared ? 1 : 0, (const uint32_t *)q_inf, (uint32_t *)f, (uint32_t)n);
    return (int)hipGetLastError();
}
int bn254_launch_gt_mul_B(const void *a, const void *b, void *out, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_gt_mul_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, (uint32_t)n);
    return (int)hipGetLastError();
}
size_t bn254_gt_pow_table_bytes_B(size_t n) {
    size_t grid = (2 * n + BLOCK - 1) / BLOCK;
    return grid * BLOCK * POW_TABLE_DWORDS_PER_LANE * sizeof(uint32_t);
}
// mode (BN254_OPT_GT_POW_MODE): 0 Frobenius decomposition for cyclotomic input, 2 strict (one-dimensional chain for cyclotomic input: no
// assumption on the order), 1 the general chain for everything
int bn254_launch_gt_pow_B(const void *a, const void *k, void *out, size_t n, void *table, int mode, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_gt_pow_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)a, (const uint32_t *)k, (uint32_t *)out, (uint32_t)n, (uint32_t *)table, mode < 0 || mode > 2 ? 0 : mode);
    return (int)hipGetLastError();
}
int bn254_launch_gt_inverse_B(const void *a, void *out, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_gt_inverse_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)a, (uint32_t *)out, (uint32_t)n);
    return (int)hipGetLastError();
}
// m pairs per lane pair on one accumulator (m = 2 or 4): ceil(n / m) values out; `state`: bn254_miller_shared_state_bytes_B(n, m)
size_t bn254_miller_shared_state_bytes_B(size_t n, int m) {
    const size_t groups = (n + m - 1) / m, grid = (2 * groups + BLOCK - 1) / BLOCK;
    return grid * BLOCK * (size_t)m * MILLER_STATE_BYTES_PER_LANE_AND_PAIR;
}
int bn254_launch_miller_shared_B(const void *p, const void *q, void *f, size_t n, int m, void *state, hipStream_t s) {
    const size_t groups = (n + m - 1) / m;
    unsigned grid = (unsigned)((2 * groups + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(m == 4 ? bn254_miller_shared4_B : bn254_miller_shared2_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint32_t *)q, (uint32_t *)f, (uint32_t)n, (uint4 *)state);
    return (int)hipGetLastError();
}
int bn254_launch_miller_B(const void *p, const void *q, void *f, size_t n, int naf, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(naf ? bn254_miller_naf_B : bn254_miller_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint32_t *)q, (uint32_t *)f, (uint32_t)n);
    return (int)hipGetLastError();
}
size_t bn254_final_exp_table_bytes_B(size_t n) {
    size_t grid = (2 * n + BLOCK - 1) / BLOCK;
    return grid * BLOCK * EXP_TABLE_DWORDS_PER_LANE * sizeof(uint32_t);
}
int bn254_launch_final_exp_B(const void *f, void *out, size_t n, void *table, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_final_exp_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)f, (uint32_t *)out, (uint32_t)n, (uint32_t *)table);
    return (int)hipGetLastError();
}
}

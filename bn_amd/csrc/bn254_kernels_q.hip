// Four-lanes-per-pairing kernels of the BN254 pairing engine for MI355X (gfx950): csrc/quad.hpp - an Fq12 split over a quad of
// lanes (lower lane pair c0, upper lane pair c1, DPP quad_perm [2,3,0,1] between them), the G2 point arithmetic of the Miller loop
// redundantly on both pairs.  The host (bn254_hip.hip) runs these for batches between BN254_OPT_WAVE_PAIRING_MAX and
// BN254_OPT_QUAD_MAX pairings per call, where the lane-pair kernels (bn254_kernels_b.hip: 4.2 ms whatever the count) would leave SIMDs
// empty: 16 384 pairings are 1024 waves here - one per SIMD - against 512 there.  Everything on the hot loops is inlined, like in
// the lane-pair kernels; with one wave per SIMD the register budget is 512 VGPRs, so nothing spills.
#define BN_INLINE_ALL 1       // fe.hpp: leaves and Fq6/Fq12-sized steps force-inlined
#include <hip/hip_runtime.h>
#include "quad.hpp"
#include "io.hpp"

using namespace bn254;

namespace {
constexpr int BLOCK = 64;
typedef Fq2B<Fe> F2;

// Miller-loop state parked in LDS between steps (R, the point being added, P): 63 dwords per lane, [dword][lane] (conflict-free) -
// the layout of bn254_kernels_b.hip's MillerStateLds; both pairs of a quad keep their own copy
constexpr int PARK_DWORDS = 63;
struct MillerStateLdsQ {
    uint32_t *base;
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

// f_out[i] = Miller value of (p[i], q[i]) on the NAF schedule (only meets a final exponentiation); infinity -> one (groups/mod.rs:766)
__global__ void __launch_bounds__(BLOCK) bn254_miller_naf_Q(const uint32_t *g1, const uint32_t *g2, uint32_t *f_out, uint32_t n) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t quad = t >> 2;
    const bool live = quad < n;
    if (!live) quad = n - 1;                                   // keep all four lanes of every quad active for the DPP exchanges
    const uint32_t *w1 = g1 + 24u * quad, *w2 = g2 + 48u * quad;
    const bool inf = words_all_zero(w1 + 16, 8) || words_all_zero(w2 + 32, 16);
    G1Aff<Fe> p;
    G2Aff<F2> q;
    pair_prologue<Fe>(f2_scalar_load((const F2 *)nullptr, w1), f2_scalar_load((const F2 *)nullptr, w1 + 8), f2_scalar_load((const F2 *)nullptr, w1 + 16),
                      f2_load((const F2 *)nullptr, w2), f2_load((const F2 *)nullptr, w2 + 16), f2_load((const F2 *)nullptr, w2 + 32), p, q);
    __shared__ uint32_t park[PARK_DWORDS * BLOCK];
    MillerStateLdsQ st = {park + threadIdx.x};
    QFq12<F2> f = q_miller_loop_naf(p, q, st);
    const QFq12<F2> one = q12_one<F2>();
    f.h.c0 = f2_select(inf, f.h.c0, one.h.c0); f.h.c1 = f2_select(inf, f.h.c1, one.h.c1); f.h.c2 = f2_select(inf, f.h.c2, one.h.c2);
    if (live) q12_store(f, f_out + 96u * quad);
}

// Table of the exponentiation machine, this lane's Fq6 halves: slot-major, 7 groups of 4 dwords per lane (27 + 1 pad), group-major then
// the lane - every access of a wave is one coalesced dwordx4 instruction over 1 KB (bn254_kernels_b.hip ExpTableMem's layout)
struct ExpTableMemQ {
    uint4 *table;
    uint32_t lane, stride;
    __device__ __forceinline__ void put(int slot, const Fq6<F2> &v) const {
        const uint32_t r = (uint32_t)(slot * 7) * stride + lane;
        uint32_t w[28];
#pragma unroll
        for (int i = 0; i < 9; ++i) { w[i] = v.c0.v.l[i]; w[9 + i] = v.c1.v.l[i]; w[18 + i] = v.c2.v.l[i]; }
        w[27] = 0;
#pragma unroll
        for (int g = 0; g < 7; ++g) table[r + (uint32_t)g * stride] = make_uint4(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]);
    }
    __device__ __forceinline__ Fq6<F2> get(int slot) const {
        const uint32_t r = (uint32_t)(slot * 7) * stride + lane;
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
};
constexpr size_t EXP_TABLE_BYTES_PER_LANE_Q = (size_t)k::EXP_SLOTS * 7 * 16;

__global__ void __launch_bounds__(BLOCK) bn254_final_exp_Q(const uint32_t *f_in, uint32_t *out, uint32_t n, uint4 *table) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t quad = t >> 2;
    const bool live = quad < n;
    if (!live) quad = n - 1;
    ExpTableMemQ tbl = {table, t, gridDim.x * BLOCK};
    const QFq12<F2> f = q_final_exponentiation(q12_load<F2>(f_in + 96u * quad), tbl);
    if (live) q12_store(f, out + 96u * quad);
}
}  // namespace

extern "C" {
int bn254_launch_miller_Q(const void *p, const void *q, void *f, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((4 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_miller_naf_Q, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint32_t *)q, (uint32_t *)f, (uint32_t)n);
    return (int)hipGetLastError();
}
size_t bn254_final_exp_table_bytes_Q(size_t n) {
    const size_t grid = (4 * n + BLOCK - 1) / BLOCK;
    return grid * BLOCK * EXP_TABLE_BYTES_PER_LANE_Q;
}
int bn254_launch_final_exp_Q(const void *f, void *out, size_t n, void *table, hipStream_t s) {
    unsigned grid = (unsigned)((4 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_final_exp_Q, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)f, (uint32_t *)out, (uint32_t)n, (uint4 *)table);
    return (int)hipGetLastError();
}
}

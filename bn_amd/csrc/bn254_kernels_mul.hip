// G * Fr kernels of the BN254 engine for MI355X (gfx950): `Mul<Fr> for G<P>` (src/groups/mod.rs:250-270) batched.
//   bn254_g1_mul_M   one lane per point (G1 is over Fq: a Jacobian point is 27 VGPRs)
//   bn254_g2_mul_M   one point per lane PAIR (Fq2B: even lane = c0, odd lane = c1 of every coordinate, DPP exchange)
// normalize = 0 runs the reference's own double-and-add chain (raw Jacobian limbs identical to the crate's, used to make
// benchmark inputs with z != 1); normalize = 1 returns the normalized point by the short chains of curve.hpp: GLV (G1) / GLS (G2)
// decomposition, signed 4-bit windows, a window table brought to a common z so that every addition is a mixed one.
//
// This translation unit inlines the point operations and the multiplier leaves: the running point stays in registers for the
// whole chain (in the call-based build of bn254_hip.hip every doubling went through private memory: 6 / 27 GB of HBM traffic
// per 2^16 G1 / G2 multiplications).  Only the 16-entry window table is a per-lane array in private memory.
#define BN_INLINE_ALL 1       // fe.hpp: leaves and Fq6/Fq12-sized steps force-inlined
#define BN_MUL_WAVES 3        // resident waves per SIMD the G1 kernels are compiled for (168 VGPRs, 7 spilled; 2: equal since round 6's single launch, 4: 72 spilled, -2.5 %: profiles/r06_ab_mul_launch_size.txt)
#include <hip/hip_runtime.h>
#include <type_traits>
// The G2 kernel runs two resident waves per SIMD like the lane-pair pairing kernels, and the same ONE hand-over of the issue priority keeps both
// alive to the end (bn254_kernels_b.hip bn_fair_handover; the hardware's oldest-first arbitration lets the older wave leave early and the younger
// one finish alone): 2^16 / 2^17 / 2^18 G2 multiplications per call 33.5 / 35.1 / 36.1 -> 35.7 / 36.6 / 36.9 M/s (profiles/r06_ab_mul_launch_size.txt).
// Called from the window loop of scalar_mul_gls only (curve.hpp BN_MUL_HOOK); the G1 kernels run three waves per SIMD and keep the default.
__device__ __forceinline__ void bn_mul_fair_handover(int step, int total) {
    const uint32_t slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 1;      // HW_ID.wave_id
    if (slot == 0) { if (step * 1000 < 769 * total) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
    else { if (step * 1000 < 231 * total) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2); }
}
#define BN_MUL_HOOK(step, total) bn_mul_fair_handover(step, total)
// The G1 kernel (three resident waves per SIMD, oldest first) ends a launch with every SIMD draining its last waves one after the other.  In the
// LAST resident round of a multi-round launch (the last 3 x 4 x 256 workgroups: MI355X; elsewhere the policy is merely mis-sized) a wave lowers
// its own priority as it advances, so the three share the SIMD by progress and finish together: +0.9 ... +1.6 % at 2^20 per launch
// (profiles/r06_ab_mul_launch_size.txt).  Single-round launches keep the default (waves of one age in lockstep are slower).
constexpr unsigned BN_G1_RESIDENT_WAVES = BN_MUL_WAVES * 4 * 256;
__device__ __forceinline__ void bn_g1_tail_policy(int step, int total) {
    if (gridDim.x >= 2 * BN_G1_RESIDENT_WAVES && blockIdx.x + BN_G1_RESIDENT_WAVES >= gridDim.x) {
        const int q = step * 4 / total;
        if (q == 0) __builtin_amdgcn_s_setprio(3); else if (q == 1) __builtin_amdgcn_s_setprio(2); else if (q == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
}
#define BN_G1_HOOK(step, total) bn_g1_tail_policy(step, total)
#include "curve.hpp"
#include "io.hpp"

using namespace bn254;

namespace {
constexpr int BLOCK = 64;
typedef Fq2B<Fe> F2;

// The affine window table of a lane in global memory: [lane][entry 1..8][18 dwords padded to 80 bytes] - a lane reads the entry of
// ITS digit as five 16-byte loads from one or two cache lines (curve.hpp AffTableVars explains why not a private array).
// 16-byte groups per entry: 5 = packed (80 B: an entry may straddle two 128-byte lines; one whole line per entry measured no faster, a
// prefetch one window ahead 2 % slower: profiles/r04c_ab_g1mul.txt)
constexpr uint32_t AFF_ENTRY_U4 = 5, AFF_LANE_U4 = 8 * AFF_ENTRY_U4;                 // 80 B per entry, 640 B per lane
template <class F>
struct AffTableMem {
    uint4 *base;             // this lane's 8 entries
    __device__ __forceinline__ static void split(const Fe &a, const Fe &b, uint32_t *w) {
#pragma unroll
        for (int i = 0; i < 9; ++i) { w[i] = a.l[i]; w[9 + i] = b.l[i]; }
        w[18] = 0; w[19] = 0;
    }
    __device__ __forceinline__ void put_fe(int i, const Fe &x, const Fe &y) const {
        uint32_t w[20];
        split(x, y, w);
        uint4 *e = base + (uint32_t)(i - 1) * AFF_ENTRY_U4;
#pragma unroll
        for (int g = 0; g < 5; ++g) e[g] = make_uint4(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]);
    }
    __device__ __forceinline__ void get_fe(int i, Fe &x, Fe &y) const {
        const uint4 *e = base + (uint32_t)(i - 1) * AFF_ENTRY_U4;
        uint32_t w[20];
#pragma unroll
        for (int g = 0; g < 5; ++g) { const uint4 v = e[g]; w[4 * g] = v.x; w[4 * g + 1] = v.y; w[4 * g + 2] = v.z; w[4 * g + 3] = v.w; }
#pragma unroll
        for (int i2 = 0; i2 < 9; ++i2) { x.l[i2] = w[i2]; y.l[i2] = w[9 + i2]; }
    }
    // G1: (x, y) are Fe; G2 in the lane-pair mapping: this lane's components of (x, y)
    __device__ __forceinline__ void put(int i, const Aff<FqField> &v) const { put_fe(i, v.x, v.y); }
    __device__ __forceinline__ void put(int i, const Aff<Fq2Field<Fq2B<Fe>>> &v) const { put_fe(i, v.x.v, v.y.v); }
    __device__ __forceinline__ Aff<F> get(int i) const {
        Aff<F> r;
        if constexpr (std::is_same<F, FqField>::value) get_fe(i, r.x, r.y);
        else get_fe(i, r.x.v, r.y.v);
        return r;
    }
};

// NORMALIZE is a template parameter, i.e. each flavour is its OWN kernel: with a run-time flag the reference chain and the GLV /
// windowed chain were register-allocated together (round 2: 74 spilled VGPRs in the G1 kernel).
template <class F, bool NORMALIZE>
__device__ __forceinline__ Jac<F> run_chain(const Jac<F> &p, const uint32_t *km, uint4 *table, uint32_t lane) {
    uint32_t kw[8], raw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kw[i] = km[i];
    fr_from_mont(kw, raw);
    if constexpr (NORMALIZE) {
#ifdef BN_AB_ALIAS_SCRATCH
        // TIMING EXPERIMENT ONLY (wrong results): all waves use the tables of the first 64 lanes (40 KB: cache resident) - the same
        // instruction stream without the HBM traffic of 640 B of table per lane
        AffTableMem<F> tab = {table + (size_t)(lane & 63u) * AFF_LANE_U4};
#else
        AffTableMem<F> tab = {table + (size_t)lane * AFF_LANE_U4};
#endif
        if constexpr (std::is_same<F, FqField>::value) return jac_normalize<F>(scalar_mul_glv(p, raw, tab));      // G1: GLV + signed windows
        else return jac_normalize<F>(scalar_mul_gls<F2>(p, raw, tab));                                             // G2: GLS, four signed-window streams
    } else {
        return scalar_mul_reference_chain<F>(p, raw);
    }
}

template <bool NORMALIZE>
__device__ __forceinline__ void g1_mul_body(const uint32_t *p, const uint32_t *k, uint32_t *out, uint32_t n, uint4 *table) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    const uint32_t *w = p + 24u * idx;
    Jac<FqField> pt = {fe_from_u32x8(w), fe_from_u32x8(w + 8), fe_from_u32x8(w + 16)};
    Jac<FqField> r = run_chain<FqField, NORMALIZE>(pt, k + 8u * idx, table, idx);
    uint32_t *o = out + 24u * idx;
    fe_to_u32x8(r.x, o); fe_to_u32x8(r.y, o + 8); fe_to_u32x8(r.z, o + 16);
}
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_MUL_WAVES, BN_MUL_WAVES))) bn254_g1_mul_M(const uint32_t *p, const uint32_t *k, uint32_t *out, uint32_t n, uint4 *table) {
    g1_mul_body<true>(p, k, out, n, table);
}
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_MUL_WAVES, BN_MUL_WAVES))) bn254_g1_mul_chain_M(const uint32_t *p, const uint32_t *k, uint32_t *out, uint32_t n) {
    g1_mul_body<false>(p, k, out, n, nullptr);
}

template <bool NORMALIZE>
__device__ __forceinline__ void g2_mul_body(const uint32_t *p, const uint32_t *k, uint32_t *out, uint32_t n, uint4 *table) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;                       // keep both lanes of every pair active for the DPP exchanges
    const uint32_t *w = p + 48u * pair;
    typedef Fq2Field<F2> F;
    Jac<F> pt = {f2_load((const F2 *)nullptr, w), f2_load((const F2 *)nullptr, w + 16), f2_load((const F2 *)nullptr, w + 32)};
    Jac<F> r = run_chain<F, NORMALIZE>(pt, k + 8u * pair, table, t);
    if (live) {
        uint32_t *o = out + 48u * pair;
        f2_store(r.x, o); f2_store(r.y, o + 16); f2_store(r.z, o + 32);
    }
}
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) bn254_g2_mul_M(const uint32_t *p, const uint32_t *k, uint32_t *out, uint32_t n, uint4 *table) {
    g2_mul_body<true>(p, k, out, n, table);
}
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) bn254_g2_mul_chain_M(const uint32_t *p, const uint32_t *k, uint32_t *out, uint32_t n) {
    g2_mul_body<false>(p, k, out, n, nullptr);
}
// a[i] + b[i]  (or a[i] - b[i] = a[i] + (-b[i]): lib.rs:103-114,146-157, groups/mod.rs:275-347): the reference's add-2007-bl
// with its zero / equal-point branches, so the Jacobian limbs returned are the reference's own
template <class F>
__device__ __forceinline__ Jac<F> add_body(const Jac<F> &a, Jac<F> b, int negate_b) {
    const bool bz = F::is_zero(b.z);
    if (negate_b) b.y = F::select(bz, F::template lc3<-1, 0, 0>(b.y, b.y, b.y), b.y);     // neg(0) = 0 (groups/mod.rs:334-346)
    return jac_add_flags<F>(a, b, F::is_zero(a.z), bz);
}
__global__ void __launch_bounds__(BLOCK) bn254_g1_add_M(const uint32_t *a, const uint32_t *b, uint32_t *out, uint32_t n, int negate_b) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    const uint32_t *wa = a + 24u * idx, *wb = b + 24u * idx;
    Jac<FqField> pa = {fe_from_u32x8(wa), fe_from_u32x8(wa + 8), fe_from_u32x8(wa + 16)};
    Jac<FqField> pb = {fe_from_u32x8(wb), fe_from_u32x8(wb + 8), fe_from_u32x8(wb + 16)};
    Jac<FqField> r = add_body<FqField>(pa, pb, negate_b);
    uint32_t *o = out + 24u * idx;
    fe_to_u32x8(r.x, o); fe_to_u32x8(r.y, o + 8); fe_to_u32x8(r.z, o + 16);
}
__global__ void __launch_bounds__(BLOCK) bn254_g2_add_M(const uint32_t *a, const uint32_t *b, uint32_t *out, uint32_t n, int negate_b) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    typedef Fq2Field<F2> F;
    const uint32_t *wa = a + 48u * pair, *wb = b + 48u * pair;
    Jac<F> pa = {f2_load((const F2 *)nullptr, wa), f2_load((const F2 *)nullptr, wa + 16), f2_load((const F2 *)nullptr, wa + 32)};
    Jac<F> pb = {f2_load((const F2 *)nullptr, wb), f2_load((const F2 *)nullptr, wb + 16), f2_load((const F2 *)nullptr, wb + 32)};
    Jac<F> r = add_body<F>(pa, pb, negate_b);
    if (live) {
        uint32_t *o = out + 48u * pair;
        f2_store(r.x, o); f2_store(r.y, o + 16); f2_store(r.z, o + 32);
    }
}
}  // namespace

extern "C" {
int bn254_launch_g1_add_M(const void *a, const void *b, void *out, size_t n, int negate_b, hipStream_t s) {
    unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_g1_add_M, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, (uint32_t)n, negate_b);
    return (int)hipGetLastError();
}
int bn254_launch_g2_add_M(const void *a, const void *b, void *out, size_t n, int negate_b, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_g2_add_M, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, (uint32_t)n, negate_b);
    return (int)hipGetLastError();
}
// `table` (normalize != 0 only): bn254_mul_table_bytes_M(g, n) bytes of scratch for the window tables of this launch
size_t bn254_mul_table_bytes_M(int g, size_t n) {
    const size_t lanes = g == 1 ? (n + BLOCK - 1) / BLOCK * BLOCK : (2 * n + BLOCK - 1) / BLOCK * BLOCK;
    return lanes * AFF_LANE_U4 * sizeof(uint4);
}
int bn254_launch_g1_mul_M(const void *p, const void *k, void *out, size_t n, int normalize, void *table, hipStream_t s) {
    unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
    if (normalize) hipLaunchKernelGGL(bn254_g1_mul_M, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint32_t *)k, (uint32_t *)out, (uint32_t)n, (uint4 *)table);
    else hipLaunchKernelGGL(bn254_g1_mul_chain_M, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint32_t *)k, (uint32_t *)out, (uint32_t)n);
    return (int)hipGetLastError();
}
int bn254_launch_g2_mul_M(const void *p, const void *k, void *out, size_t n, int normalize, void *table, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    if (normalize) hipLaunchKernelGGL(bn254_g2_mul_M, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint32_t *)k, (uint32_t *)out, (uint32_t)n, (uint4 *)table);
    else hipLaunchKernelGGL(bn254_g2_mul_chain_M, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint32_t *)k, (uint32_t *)out, (uint32_t)n);
    return (int)hipGetLastError();
}
}

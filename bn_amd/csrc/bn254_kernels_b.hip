// Lane-pair mapping (Fq2B) kernels of the BN254 pairing engine for MI355X (gfx950): TWO adjacent lanes compute one
// pairing, the even lane holding the real and the odd lane the imaginary component of every Fq2 value.  Per-lane state
// halves (an Fq12 is 54 VGPRs), so the Miller loop and the exponentiation by u run out of registers instead of private
// memory; the partner's limbs arrive by DPP (quad_perm [1,0,3,2]).  A batch of 2^16 pairings is 2048 waves = 2 per SIMD,
// which is what the integer pipe needs to be saturated (profiles/r01_ubench_valu_rates.txt).
//
// This translation unit is compiled with the Fq6/Fq12-sized steps INLINED (BN_COARSE), so that values stay in VGPRs across
// them; only the multiplier-sized leaves and the rarely executed outer steps are calls.
#define BN_COARSE __device__ __forceinline__
#ifndef BN_WAVES
#define BN_WAVES 2          // resident waves per SIMD the kernels are compiled for (256 VGPRs each)
#endif
// ... and, since the measurement of round 1f, the multiplier- and reduction-sized leaves as well: without calls there is no
// argument marshalling (v_mov was ~45 % of the non-leaf instructions) and no caller-saved/callee-saved split of the register
// file.  The Miller loop body becomes ~160 KB of straight-line code (beyond the 64 KB instruction cache), and is still 7 %
// faster than the call-based build (profiles/r01g_*).  -DBN_B_CALL_LEAVES restores the calls.
#ifndef BN_B_CALL_LEAVES
#define BN_LEAF_MUL __device__ __forceinline__
#define BN_LEAF_RED __device__ __forceinline__
#endif
#ifdef BN_B_CALL_MUL               // experiments: call only one of the two leaf classes
#undef BN_LEAF_MUL
#endif
#ifdef BN_B_CALL_RED
#undef BN_LEAF_RED
#endif
#ifdef BN_B_INLINE_REDUCTIONS
#define BN_INLINE_REDUCTIONS 1
#endif
#ifndef BN_B_BLOCK
#define BN_B_BLOCK 64
#endif
#if BN_B_BLOCK > 64
// several waves per workgroup: re-align them at every loop step so that they walk the (larger than the instruction cache)
// loop bodies together and share the instruction fetches
#define BN_LOOP_SYNC() __syncthreads()
#endif
#include <hip/hip_runtime.h>
#include "curve.hpp"
#include "io.hpp"

using namespace bn254;

namespace {
constexpr int BLOCK = BN_B_BLOCK;
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
    Fq12<F2> f = miller_loop_sched<NAF>(p, q, st);
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

// Table of odd powers of the windowed exponentiation (pairing.hpp ExpTableVars) in global memory: slot-major, then the 54
// dwords of this lane's half of an Fq12, then the lane - every access of a wave is one coalesced 256-byte row.
struct ExpTableMem {
    uint32_t *table;         // wave-uniform base (SGPRs)
    uint32_t lane;           // this lane's column
    uint32_t stride;         // lanes in the launch
    __device__ __forceinline__ uint32_t *row(int slot, int half) const {          // uniform: scalar address arithmetic
        return table + (size_t)(uint32_t)((slot * 2 + half) * 27) * stride;
    }
    __device__ __forceinline__ void st6(int slot, int half, const Fq6<F2> &v) const {
        uint32_t *p = row(slot, half);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            (p + (size_t)i * stride)[lane] = v.c0.v.l[i];
            (p + (size_t)(9 + i) * stride)[lane] = v.c1.v.l[i];
            (p + (size_t)(18 + i) * stride)[lane] = v.c2.v.l[i];
        }
    }
    __device__ __forceinline__ Fq6<F2> ld6(int slot, int half) const {
        const uint32_t *p = row(slot, half);
        Fq6<F2> v;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            v.c0.v.l[i] = (p + (size_t)i * stride)[lane];
            v.c1.v.l[i] = (p + (size_t)(9 + i) * stride)[lane];
            v.c2.v.l[i] = (p + (size_t)(18 + i) * stride)[lane];
        }
        return v;
    }
    __device__ __forceinline__ void put(int i, const Fq12<F2> &v) const { st6(i, 0, v.c0); st6(i, 1, v.c1); }
    __device__ __forceinline__ Fq6<F2> c0(int i) const { return ld6(i, 0); }
    __device__ __forceinline__ Fq6<F2> c1(int i) const { return ld6(i, 1); }
};
constexpr size_t EXP_TABLE_DWORDS_PER_LANE = (size_t)k::EXP_SLOTS * 54;

__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_final_exp_B(const uint32_t *f_in, uint32_t *out, uint32_t n, uint32_t *table) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    ExpTableMem tbl = {table, t, gridDim.x * BLOCK};
    Fq12<F2> f = final_exponentiation(f12_load<F2>(f_in + 96u * pair), tbl);
    if (live) f12_store(f, out + 96u * pair);
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
    Fq12<F2> f = miller_loop_prepared<F2>(p, source);
    Fq12<F2> one = f12_one<F2>();
    f.c0.c0 = f2_select(inf, f.c0.c0, one.c0.c0); f.c0.c1 = f2_select(inf, f.c0.c1, one.c0.c1); f.c0.c2 = f2_select(inf, f.c0.c2, one.c0.c2);
    f.c1.c0 = f2_select(inf, f.c1.c0, one.c1.c0); f.c1.c1 = f2_select(inf, f.c1.c1, one.c1.c1); f.c1.c2 = f2_select(inf, f.c1.c2, one.c1.c2);
    if (live) f12_store(f, f_out + 96u * pair);
}

// out[t] = product of in[t*chunk .. min(n, (t+1)*chunk))   (product tree of the multi-pairing; chunk is small so that every
// level keeps many lane pairs busy: 2^15 values -> 1 in 8 levels of 3 multiplications instead of 3 levels of 63)
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_gt_product_B(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t chunk) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    uint32_t groups = (n + chunk - 1) / chunk;
    bool live = pair < groups;
    if (!live) pair = groups - 1;
    uint32_t lo = pair * chunk, hi = lo + chunk < n ? lo + chunk : n;
    Fq12<F2> acc = f12_load<F2>(in + 96u * lo);
#pragma unroll 1
    for (uint32_t j = lo + 1; j < hi; ++j) acc = f12_mul_o(acc, f12_load<F2>(in + 96u * j));
    if (live) f12_store(acc, out + 96u * pair);
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
// taken out of Montgomery form).  Exponent bits differ per element, so the conditional product is a per-pair select.
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(BN_WAVES, BN_WAVES))) bn254_gt_pow_B(const uint32_t *a, const uint32_t *k, uint32_t *out, uint32_t n) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    uint32_t kw[8], raw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kw[i] = k[8u * pair + i];
    fr_from_mont(kw, raw);
    Fq12<F2> base = f12_load<F2>(a + 96u * pair);
    Fq12<F2> res = f12_one<F2>();
#pragma unroll 1
    for (int i = 255; i >= 0; --i) {
        res = f12_sqr(res);
        Fq12<F2> m = f12_mul(base, res);
        bool bit = (raw[i >> 5] >> (i & 31)) & 1;
        res.c0.c0 = f2_select(bit, res.c0.c0, m.c0.c0); res.c0.c1 = f2_select(bit, res.c0.c1, m.c0.c1); res.c0.c2 = f2_select(bit, res.c0.c2, m.c0.c2);
        res.c1.c0 = f2_select(bit, res.c1.c0, m.c1.c0); res.c1.c1 = f2_select(bit, res.c1.c1, m.c1.c1); res.c1.c2 = f2_select(bit, res.c1.c2, m.c1.c2);
    }
    if (live) f12_store(res, out + 96u * pair);
}
}  // namespace

extern "C" {
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
int bn254_launch_gt_product_B(const void *in, void *out, size_t n, unsigned chunk, hipStream_t s) {
    size_t groups = (n + chunk - 1) / chunk;
    unsigned grid = (unsigned)((2 * groups + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_gt_product_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)in, (uint32_t *)out, (uint32_t)n, chunk);
    return (int)hipGetLastError();
}
int bn254_launch_gt_mul_B(const void *a, const void *b, void *out, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_gt_mul_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)a, (const uint32_t *)b, (uint32_t *)out, (uint32_t)n);
    return (int)hipGetLastError();
}
int bn254_launch_gt_pow_B(const void *a, const void *k, void *out, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_gt_pow_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)a, (const uint32_t *)k, (uint32_t *)out, (uint32_t)n);
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

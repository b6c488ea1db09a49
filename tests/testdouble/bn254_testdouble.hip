// TEST INFRASTRUCTURE, not part of the product: the ONE-LANE-PER-PAIRING mapping of the engine (Fq2A of bn_amd/csrc/fq2.hpp: Karatsuba Fq2, one
// lane holds both components, no cross-lane traffic; the Fq6/Fq12-sized steps are calls whose operands travel through private memory) as a
// small library of its own.  Rounds 1-4 shipped these kernels inside libbn254_hip.so behind bn254_ctx_set_mapping(ctx, 0); they were never
// a performance path (329 VGPRs, 18 spills, 3-8 KB of private segment: profiles/r01a_*) and their one user was the test suite, which runs
// them as a SECOND implementation of the same templates (tower.hpp / pairing.hpp / curve.hpp over a different Fq2) at sizes the CPU oracle
// cannot cover - every pairing of a 2^16 or 2^17 batch compared bit for bit.  Round 5 moved them here:
// tests/testdouble.py builds this file next to the product library's headers and tests/test_gpu_parity.py drives it.
//   reference semantics: pairing /root/reference/src/groups/mod.rs:764-771, miller_loop :486-519, final_exponentiation fields/fq12.rs:41-88,
//   G * Fr :250-270 with normalize() of lib.rs:88-95
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>

#include "curve.hpp"
#include "io.hpp"

using namespace bn254;

namespace {
constexpr int BLOCK = 64;

template <class F2>
__device__ __forceinline__ void miller_body(const uint32_t *__restrict__ g1, const uint32_t *__restrict__ g2, uint32_t *__restrict__ f_out) {
    uint32_t w1[24], w2[48];
#pragma unroll
    for (int i = 0; i < 24; ++i) w1[i] = g1[i];
#pragma unroll
    for (int i = 0; i < 48; ++i) w2[i] = g2[i];
    bool inf = words_all_zero(w1 + 16, 8) || words_all_zero(w2 + 32, 16);        // groups/mod.rs:766
    G1Aff<Fe> p = g1_to_affine(fe_from_u32x8(w1), fe_from_u32x8(w1 + 8), fe_from_u32x8(w1 + 16));
    G2Aff<F2> q = g2_to_affine(f2_load((const F2 *)nullptr, w2), f2_load((const F2 *)nullptr, w2 + 16), f2_load((const F2 *)nullptr, w2 + 32));
    Fq12<F2> f = miller_loop(p, q);                                               // the reference's schedule: the value is the reference's
    uint32_t o[96];
    f12_store(f, o);
    uint32_t one[8];                                                              // Gt::one(): c0.c0.c0 = R mod q, everything else 0
    fe_to_u32x8(fe_one(), one);
#pragma unroll
    for (int i = 0; i < 96; ++i) f_out[i] = inf ? (i < 8 ? one[i] : 0u) : o[i];
}
__global__ void __launch_bounds__(BLOCK) bntd_miller_A(const uint32_t *g1, const uint32_t *g2, uint32_t *f_out, uint32_t n) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    miller_body<Fq2A>(g1 + 24u * idx, g2 + 48u * idx, f_out + 96u * idx);
}
__global__ void __launch_bounds__(BLOCK) bntd_final_exp_A(const uint32_t *f_in, uint32_t *out, uint32_t n) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    uint32_t w[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) w[i] = f_in[96u * idx + i];
    Fq12<Fq2A> f = final_exponentiation(f12_load<Fq2A>(w));
    f12_store(f, w);
#pragma unroll
    for (int i = 0; i < 96; ++i) out[96u * idx + i] = w[i];
}
// out[t] = product of in[t*chunk .. min(n,(t+1)*chunk))   (one Fq12 chain per lane)
__global__ void __launch_bounds__(BLOCK) bntd_gt_product_A(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t chunk) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint64_t lo = (uint64_t)t * chunk;
    if (lo >= n) return;
    uint64_t hi = lo + chunk < n ? lo + chunk : n;
    uint32_t w[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) w[i] = in[96u * lo + i];
    Fq12<Fq2A> acc = f12_load<Fq2A>(w);
    for (uint64_t j = lo + 1; j < hi; ++j) {
#pragma unroll
        for (int i = 0; i < 96; ++i) w[i] = in[96u * j + i];
        acc = f12_mul(acc, f12_load<Fq2A>(w));
    }
    f12_store(acc, w);
#pragma unroll
    for (int i = 0; i < 96; ++i) out[96u * t + i] = w[i];
}
template <class F, int W, class LD, class ST>
__device__ __forceinline__ void mul_body(const uint32_t *pt, const uint32_t *km, uint32_t *out, int normalize, LD ld, ST st) {
    uint32_t kw[8], raw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kw[i] = km[i];
    fr_from_mont(kw, raw);
    uint32_t w[3 * W];
#pragma unroll
    for (int i = 0; i < 3 * W; ++i) w[i] = pt[i];
    Jac<F> p = {ld(w), ld(w + W), ld(w + 2 * W)};
    // normalize = 0: the reference's own chain (raw Jacobian limbs, bit-identical to `G * Fr`); 1: plain 4-bit windows + normalized
    Jac<F> r = normalize ? jac_normalize<F>(scalar_mul_windowed<F>(p, raw)) : scalar_mul_reference_chain<F>(p, raw);
    st(r.x, w); st(r.y, w + W); st(r.z, w + 2 * W);
#pragma unroll
    for (int i = 0; i < 3 * W; ++i) out[i] = w[i];
}
__global__ void __launch_bounds__(BLOCK) bntd_g1_mul_k(const uint32_t *p, const uint32_t *k, uint32_t *out, uint32_t n, int normalize) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    mul_body<FqField, 8>(p + 24u * idx, k + 8u * idx, out + 24u * idx, normalize,
                         [](const uint32_t *w) { return fe_from_u32x8(w); }, [](const Fe &a, uint32_t *w) { fe_to_u32x8(a, w); });
}
__global__ void __launch_bounds__(BLOCK) bntd_g2_mul_k(const uint32_t *p, const uint32_t *k, uint32_t *out, uint32_t n, int normalize) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    mul_body<Fq2Field<Fq2A>, 16>(p + 48u * idx, k + 8u * idx, out + 48u * idx, normalize,
                                 [](const uint32_t *w) { return f2_load((const Fq2A *)nullptr, w); }, [](const Fq2A &a, uint32_t *w) { f2_store(a, w); });
}
inline unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }
constexpr size_t MAX_N = 0x7fffffffu / 96;            // 32-bit word offsets inside a launch
}  // namespace

// device pointers in, asynchronous on `stream`; 0 or a hipError_t
extern "C" {
int bntd_miller(const void *d_p, const void *d_q, void *d_f, size_t n, void *stream) {
    if (n == 0) return 0;
    if (n > MAX_N) return -2;
    hipLaunchKernelGGL(bntd_miller_A, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, (const uint32_t *)d_p, (const uint32_t *)d_q, (uint32_t *)d_f, (uint32_t)n);
    return (int)hipGetLastError();
}
int bntd_final_exp(const void *d_f, void *d_out, size_t n, void *stream) {
    if (n == 0) return 0;
    if (n > MAX_N) return -2;
    hipLaunchKernelGGL(bntd_final_exp_A, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, (const uint32_t *)d_f, (uint32_t *)d_out, (uint32_t)n);
    return (int)hipGetLastError();
}
// d_out[0] = product of d_in[0..n); d_tmp: 2 x ceil(n / 4) x 384 bytes
int bntd_gt_product(const void *d_in, size_t n, void *d_out, void *d_tmp, void *stream) {
    if (n == 0 || n > MAX_N) return -2;
    const uint32_t chunk = 4;
    const uint32_t *src = (const uint32_t *)d_in;
    const size_t cap = (n + chunk - 1) / chunk;
    uint32_t *bufA = (uint32_t *)d_tmp, *bufB = (uint32_t *)d_tmp + 96 * cap;
    bool useA = true;
    while (true) {
        const size_t m = (n + chunk - 1) / chunk;
        uint32_t *dst = m == 1 ? (uint32_t *)d_out : (useA ? bufA : bufB);
        hipLaunchKernelGGL(bntd_gt_product_A, dim3(grid_for(m)), dim3(BLOCK), 0, (hipStream_t)stream, src, dst, (uint32_t)n, chunk);
        int rc = (int)hipGetLastError();
        if (rc) return rc;
        if (m == 1) return 0;
        src = dst; n = m; useA = !useA;
    }
}
int bntd_g1_mul(const void *d_p, const void *d_k, void *d_out, size_t n, int normalize, void *stream) {
    if (n == 0) return 0;
    if (n > MAX_N) return -2;
    hipLaunchKernelGGL(bntd_g1_mul_k, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, (const uint32_t *)d_p, (const uint32_t *)d_k, (uint32_t *)d_out, (uint32_t)n, normalize);
    return (int)hipGetLastError();
}
int bntd_g2_mul(const void *d_p, const void *d_k, void *d_out, size_t n, int normalize, void *stream) {
    if (n == 0) return 0;
    if (n > MAX_N) return -2;
    hipLaunchKernelGGL(bntd_g2_mul_k, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, (const uint32_t *)d_p, (const uint32_t *)d_k, (uint32_t *)d_out, (uint32_t)n, normalize);
    return (int)hipGetLastError();
}
}

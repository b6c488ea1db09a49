// Measurement and synthetic-input kernels of the BN254 engine (include/bn254_hip.h "measurement" / "synthetic benchmark inputs"): nothing on
// the pairing path calls into this unit.
//   * bn254_ubench_mac32(_ex): the v_mad_u64_u32 issue-rate microbenchmark behind bench.py's same-run `roofline.peak` (tools/ubench.hip is the
//     long form; the multiplier's rate depends on its DATA - profiles/r04_ubench_mad_data_dependence.txt -, hence the operand-width argument:
//     32 random bits for `peak`, the engine's own 29-bit limbs for `peak_at_kernel_occupancy`);
//   * bn254_synthetic_scalars_dev: the on-device generator of the synthetic Fr scalars (SplitMix64 -> 512 bits -> mod r -> Montgomery form),
//     word for word bn_amd.distributed.synthetic_scalars;
//   * bn254_tile_dev: one record repeated n times (the generator bases the scalars multiply).
#include <cstdint>
#include <mutex>

#include "host_ctx.hpp"
#include "bn254_constants.hpp"

namespace {

// 16 independent-ish v_mad_u64_u32 per iteration on 8 accumulators: the issue-rate ceiling of the instruction every field
// multiplication of the engine is built from (tools/ubench.hip is the long form of this experiment)
__global__ void __launch_bounds__(256) bn254_ubench_mad_k(uint32_t *out, uint32_t seed, int iters, uint32_t operand_mask) {
    uint32_t a = (threadIdx.x * 2654435761u + seed) & operand_mask, b = (a ^ 0x9e3779b9u) & operand_mask;
    uint64_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u & 7]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0x1234567) out[threadIdx.x] = (uint32_t)s;
}

// ---- Fr (8 x u32, Montgomery radix 2^256) just for the scalar generator: CIOS product, result < r
__device__ void fr_mont_mul(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    using namespace bn254;
    uint32_t t[10] = {};
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
        for (int j = 0; j < 8; ++j) { uint64_t x = (uint64_t)a[j] * b[i] + t[j] + c; t[j] = (uint32_t)x; c = x >> 32; }
        uint64_t x = (uint64_t)t[8] + c; t[8] = (uint32_t)x; t[9] = (uint32_t)(x >> 32);
        uint32_t mq = t[0] * k::FR_INV32;
        c = ((uint64_t)mq * k::FR_MOD32[0] + t[0]) >> 32;
        for (int j = 1; j < 8; ++j) { uint64_t y = (uint64_t)mq * k::FR_MOD32[j] + t[j] + c; t[j - 1] = (uint32_t)y; c = y >> 32; }
        x = (uint64_t)t[8] + c; t[7] = (uint32_t)x;
        t[8] = t[9] + (uint32_t)(x >> 32);
        t[9] = 0;
    }
    // t < 2r: one conditional subtraction
    uint32_t d[8];
    int64_t br = 0;
    for (int i = 0; i < 8; ++i) { int64_t s = (int64_t)t[i] - (int64_t)k::FR_MOD32[i] + br; d[i] = (uint32_t)s; br = s >> 32; }
    const bool ge = (t[8] != 0) || (br == 0);
    for (int i = 0; i < 8; ++i) out[i] = ge ? d[i] : t[i];
}
__device__ uint64_t splitmix64_next(uint64_t &state) {
    state += 0x9E3779B97F4A7C15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// out[j] = Montgomery image of (512-bit SplitMix64 draw of stream 2*(lo+j)+which) mod r   (bn_amd.distributed.synthetic_scalars)
__global__ void __launch_bounds__(64) bn254_synthetic_scalars_k(uint64_t lo, uint32_t n, uint32_t which, uint64_t seed, uint32_t *out) {
    using namespace bn254;
    const uint32_t j = blockIdx.x * 64 + threadIdx.x;
    if (j >= n) return;
    uint64_t state = seed + (((lo + j) * 2 + which) << 32);
    uint32_t w[16];
    for (int i = 0; i < 8; ++i) { uint64_t z = splitmix64_next(state); w[2 * i] = (uint32_t)z; w[2 * i + 1] = (uint32_t)(z >> 32); }
    uint32_t r2[8], r3[8], a[8], b[8];
    for (int i = 0; i < 8; ++i) r2[i] = k::FR_R2_32[i];
    fr_mont_mul(r2, r2, r3);                    // R^3 mod r
    fr_mont_mul(w, r2, a);                      // low half  * R     (operand < 2^256, result < r)
    fr_mont_mul(w + 8, r3, b);                  // high half * R^2 = high * 2^256 * R
    uint32_t s[9];
    uint64_t c = 0;
    for (int i = 0; i < 8; ++i) { uint64_t x = (uint64_t)a[i] + b[i] + c; s[i] = (uint32_t)x; c = x >> 32; }
    s[8] = (uint32_t)c;
    uint32_t d[8];
    int64_t br = 0;
    for (int i = 0; i < 8; ++i) { int64_t t = (int64_t)s[i] - (int64_t)k::FR_MOD32[i] + br; d[i] = (uint32_t)t; br = t >> 32; }
    const bool ge = (s[8] != 0) || (br == 0);
    for (int i = 0; i < 8; ++i) out[8u * j + i] = ge ? d[i] : s[i];
}
// out[i] = src[0]  (tiles one point/record of `words` u32 over n records: the generator bases of the synthetic inputs)
__global__ void __launch_bounds__(256) bn254_tile_k(const uint32_t *src, uint32_t words, uint64_t total, uint32_t *out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = src[i % words];
}

}  // namespace

extern "C" {

// G MAC32/s (lane multiply-accumulates per second) of a pure v_mad_u64_u32 stream at `waves_per_simd` resident waves on operands of
// `operand_bits` random bits (32: any words; 29: the engine's limbs - the instruction is ~3 % faster on them)
int bn254_ubench_mac32_ex(bn254_ctx *ctx, int waves_per_simd, int iters, int operand_bits, double *gmac_per_s, double *ms_out) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || operand_bits < 1 || operand_bits > 32 || !gmac_per_s) return BN254_E_BAD_ARG;
    const uint32_t mask = operand_bits == 32 ? 0xffffffffu : ((1u << operand_bits) - 1u);
    std::lock_guard<std::mutex> lk(ctx->mu);
    BnDeviceGuard dev_guard;
    HIP_TRY(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    const int blocks = prop.multiProcessorCount * waves_per_simd;          // 256 threads = one wave on each of a CU's 4 SIMDs
    if ((rc = ctx->stage[0].reserve(4096))) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(bn254_ubench_mad_k, dim3(blocks), dim3(256), 0, ctx->stream, (uint32_t *)ctx->stage[0].p, 1u, iters / 8 + 1, mask);   // warm-up
    HIP_TRY(hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL(bn254_ubench_mad_k, dim3(blocks), dim3(256), 0, ctx->stream, (uint32_t *)ctx->stage[0].p, 2u, iters, mask);
    HIP_TRY(hipEventRecord(e1, ctx->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    *gmac_per_s = (double)blocks * 256.0 * 16.0 * iters / (ms * 1e-3) / 1e9;
    if (ms_out) *ms_out = ms;
    return BN254_OK;
}
int bn254_ubench_mac32(bn254_ctx *ctx, int waves_per_simd, int iters, double *gmac_per_s, double *ms_out) {
    return bn254_ubench_mac32_ex(ctx, waves_per_simd, iters, 32, gmac_per_s, ms_out);
}

int bn254_synthetic_scalars_dev(bn254_ctx *ctx, uint64_t seed, uint64_t lo, size_t n, int which, void *d_out, void *stream) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (n == 0) return BN254_OK;
    if (!d_out || n > 0x7fffffffu / 8 || (which != 0 && which != 1)) return BN254_E_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(bn254_synthetic_scalars_k, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, lo, (uint32_t)n, (uint32_t)which, seed, (uint32_t *)d_out);
    return (int)hipGetLastError();
}
int bn254_tile_dev(bn254_ctx *ctx, const void *d_record, size_t record_bytes, size_t n, void *d_out, void *stream) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (n == 0) return BN254_OK;
    if (!d_record || !d_out || record_bytes == 0 || record_bytes % 4) return BN254_E_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t total = (uint64_t)n * (record_bytes / 4);
    hipLaunchKernelGGL(bn254_tile_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)d_record, (uint32_t)(record_bytes / 4), total, (uint32_t *)d_out);
    return (int)hipGetLastError();
}

}  // extern "C"

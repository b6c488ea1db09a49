// Lane-pair mapping (Fq2B) kernels of the BN254 pairing engine for MI355X (gfx950): TWO adjacent lanes compute one
// pairing, the even lane holding the real and the odd lane the imaginary component of every Fq2 value.  Per-lane state
// halves (an Fq12 is 54 VGPRs), so the Miller loop and the exponentiation by u run out of registers instead of private
// memory; the partner's limbs arrive by DPP (quad_perm [1,0,3,2]).  A batch of 2^16 pairings is 2048 waves = 2 per SIMD,
// which is what the integer pipe needs to be saturated (profiles/r01_ubench_valu_rates.txt).
//
// This translation unit is compiled with the Fq6/Fq12-sized steps INLINED (BN_COARSE), so that values stay in VGPRs across
// them; only the multiplier-sized leaves and the rarely executed outer steps are calls.
#define BN_COARSE __device__ __forceinline__
#ifdef BN_B_INLINE_REDUCTIONS
#define BN_INLINE_REDUCTIONS 1
#endif
#include <hip/hip_runtime.h>
#include "io.hpp"

using namespace bn254;

namespace {
constexpr int BLOCK = 64;
typedef Fq2B<Fe> F2;

__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) bn254_miller_B(const uint32_t *g1, const uint32_t *g2, uint32_t *f_out, uint32_t n) {
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
    Fq12<F2> f = miller_loop(p, q);
    Fq12<F2> one = f12_one<F2>();
    f.c0.c0 = f2_select(inf, f.c0.c0, one.c0.c0); f.c0.c1 = f2_select(inf, f.c0.c1, one.c0.c1); f.c0.c2 = f2_select(inf, f.c0.c2, one.c0.c2);
    f.c1.c0 = f2_select(inf, f.c1.c0, one.c1.c0); f.c1.c1 = f2_select(inf, f.c1.c1, one.c1.c1); f.c1.c2 = f2_select(inf, f.c1.c2, one.c1.c2);
    if (live) f12_store(f, f_out + 96u * pair);
}

__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) bn254_final_exp_B(const uint32_t *f_in, uint32_t *out, uint32_t n) {
    uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t pair = t >> 1;
    bool live = pair < n;
    if (!live) pair = n - 1;
    Fq12<F2> f = final_exponentiation(f12_load<F2>(f_in + 96u * pair));
    if (live) f12_store(f, out + 96u * pair);
}
}  // namespace

extern "C" {
int bn254_launch_miller_B(const void *p, const void *q, void *f, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_miller_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)p, (const uint32_t *)q, (uint32_t *)f, (uint32_t)n);
    return (int)hipGetLastError();
}
int bn254_launch_final_exp_B(const void *f, void *out, size_t n, hipStream_t s) {
    unsigned grid = (unsigned)((2 * n + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(bn254_final_exp_B, dim3(grid), dim3(BLOCK), 0, s, (const uint32_t *)f, (uint32_t *)out, (uint32_t)n);
    return (int)hipGetLastError();
}
}

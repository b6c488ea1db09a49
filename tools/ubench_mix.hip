// Does a third (or fourth) resident wave per SIMD pay for the REAL instruction mix of the pairing kernels?  
// The stream is not synthetic: each iteration is the engine's own Granger-Scott cyclotomic squaring (tower.hpp f12_cyclotomic_sqr over the
// lane-pair Fq2 of fq2.hpp - six dual products with their DPP operand set-up, nine fused reductions: 55 % v_mad_u64_u32, 13 % v_mad_i64_i32,
// 7 % DPP moves, the rest masks / shifts / limb adds), i.e. 55 % of the final exponentiation's instructions, on field elements (29-bit limbs).
// One kernel per occupancy (amdgpu_waves_per_eu(N, N) bounds the register allocation: the VGPR / spill counts are printed by the build line
// below), launched with exactly N waves per SIMD on every CU; reports wall time per squaring, squarings per second and SIMD, and the
// effective clock (s_memtime ticks / wall time).  A second stream ("dense") is the Fq12 product f <- f * g (tower.hpp f12_mul).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibn_amd/csrc -Rpass-analysis=kernel-resource-usage tools/ubench_mix.hip -o /tmp/ubench_mix
#define BN_INLINE_ALL 1
#include <hip/hip_runtime.h>
#include "tower.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace bn254;
typedef Fq2B<Fe> F2;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ Fq12<F2> load12(const uint32_t *src) {
    Fq12<F2> f; F2 *c = &f.c0.c0;
    for (int k = 0; k < 6; ++k) for (int i = 0; i < 9; ++i) c[k].v.l[i] = src[(k * 9 + i) * 64 + (threadIdx.x & 63)] & (i < 8 ? MASK29 : 0x3fffffu);
    return f;
}
template <int WAVES, int KIND>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k_stream(const uint32_t *src, uint32_t *out, int iters) {
    Fq12<F2> f = load12(src);
    // the multiplier of the dense product is read by halves when it is needed, as the exponentiation machine reads its table (L2 hits here)
    struct GSrc { const uint32_t *p; __device__ __forceinline__ Fq6<F2> c0() const { return load12(p).c0; } __device__ __forceinline__ Fq6<F2> c1() const { return load12(p).c1; } };
    const uint32_t *gp = src + 54 * 64;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) f = f12_cyclotomic_sqr(f);
        else { asm volatile("" : "+s"(gp) :: "memory"); f = f12_mul_src(f, GSrc{gp}, false); }      // (the laundered pointer keeps the loads inside the loop)
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0; const F2 *c = &f.c0.c0;
    for (int k = 0; k < 6; ++k) for (int i = 0; i < 9; ++i) s ^= c[k].v.l[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[(size_t)gridDim.x * 256 + blockIdx.x * 4 + (threadIdx.x >> 6)] = (uint32_t)(t1 - t0);
}
typedef void (*kern_t)(const uint32_t *, uint32_t *, int);
int main(int argc, char **argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    printf("device %s  CUs %d  nominal clock %d kHz   iterations per wave %d\n", prop.name, cus, prop.clockRate, iters);
    std::vector<uint32_t> h(108 * 64);
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (auto &w : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (uint32_t)x; }
    uint32_t *src, *out; CK(hipMalloc(&src, h.size() * 4)); CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const size_t out_words = (size_t)cus * 8 * 256 + (size_t)cus * 8 * 4;
    CK(hipMalloc(&out, out_words * 4));
    std::vector<uint32_t> ho(out_words);
    struct { const char *name; int waves; kern_t k; int instr; } ks[] = {
        {"cyclotomic squaring", 1, k_stream<2, 0>, 0}, {"cyclotomic squaring", 2, k_stream<2, 0>, 0}, {"cyclotomic squaring", 3, k_stream<3, 0>, 0}, {"cyclotomic squaring", 4, k_stream<4, 0>, 0},
        {"dense Fq12 product", 1, k_stream<2, 1>, 0}, {"dense Fq12 product", 2, k_stream<2, 1>, 0}, {"dense Fq12 product", 3, k_stream<3, 1>, 0}, {"dense Fq12 product", 4, k_stream<4, 1>, 0}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-22s %6s %12s %14s %16s %8s\n", "stream", "w/SIMD", "us per op", "ops/s per SIMD", "rel. to 2 waves", "GHz");
    double base[2] = {0, 0};
    for (int rep = 0; rep < 2; ++rep)
    for (auto &b : ks) {
        const int blocks = cus * b.waves;           // 256 threads = 4 waves = one wave per SIMD per workgroup
        b.k<<<blocks, 256>>>(src, out, iters / 10 + 1); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        b.k<<<blocks, 256>>>(src, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(ho.data(), out, ((size_t)blocks * 256 + (size_t)blocks * 4) * 4, hipMemcpyDeviceToHost));
        double ticks = 0; for (int w = 0; w < blocks * 4; ++w) ticks += ho[(size_t)blocks * 256 + w];
        ticks /= blocks * 4;
        const double ops_per_simd = (double)b.waves * iters / (ms * 1e-3);
        const int kind = b.name[0] == 'c' ? 0 : 1;
        if (b.waves == 2) base[kind] = ops_per_simd;
        if (rep == 1) printf("%-22s %6d %12.3f %14.0f %16.3f %8.2f\n", b.name, b.waves, ms * 1e3 / iters, ops_per_simd, base[kind] > 0 ? ops_per_simd / base[kind] : 0.0, ticks / (ms * 1e-3) / 1e9);
    }
    return 0;
}

// Does the rate of v_mad_u64_u32 depend on the DATA?  (the chip is power-limited on this instruction: DESIGN.md section 5)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_data.hip -o tools/ubench_data && tools/ubench_data
// Patterns for the two 32-bit operands and the start of the accumulators: random 32-bit, random 29-bit (the engine's limbs), zero,
// all ones, sign-extended small negatives through v_mad_i64_i32 (the signed Karatsuba leaves).  Reported per occupancy: G mads/s/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int SIGNED>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t mask_a, uint32_t or_a, uint32_t mask_b, uint32_t or_b, int iters) {
    uint32_t a = ((threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u) & mask_a) | or_a, b = (((a ^ 0x9e3779b9u) * 2246822519u) & mask_b) | or_b;
    uint64_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (uint64_t)(a & mask_b) * (i + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (SIGNED) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[u & 7]) : "v"(a), "v"(b) : "vcc");
            else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u & 7]) : "v"(a), "v"(b) : "vcc");
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    if (s == 0x1234567) out[threadIdx.x] = (uint32_t)s;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    uint32_t *d; hipMalloc(&d, 4096);
    struct Pat { const char *name; int sgn; uint32_t ma, oa, mb, ob; } pats[] = {
        {"u64 random 32-bit x random 32-bit", 0, 0xffffffffu, 0, 0xffffffffu, 0}, {"u64 random 29-bit x random 29-bit", 0, 0x1fffffffu, 0, 0x1fffffffu, 0},
        {"u64 random 16-bit x random 16-bit", 0, 0xffffu, 0, 0xffffu, 0}, {"u64 zero x zero", 0, 0, 0, 0, 0}, {"u64 ones x ones", 0, 0, 0xffffffffu, 0, 0xffffffffu},
        {"i64 random 29-bit x random 29-bit", 1, 0x1fffffffu, 0, 0x1fffffffu, 0}, {"i64 NEGATIVE 29-bit x random 29-bit", 1, 0x1fffffffu, 0xe0000000u, 0x1fffffffu, 0},
        {"i64 negative x negative", 1, 0x1fffffffu, 0xe0000000u, 0x1fffffffu, 0xe0000000u}};
    const int iters = 20000;
    std::printf("%-40s %10s %10s %10s\n", "pattern (G mads/s per SIMD)", "1 wave", "2 waves", "8 waves");
    for (auto &q : pats) {
        std::printf("%-40s", q.name);
        for (int w : {1, 2, 8}) {
            const int blocks = p.multiProcessorCount * w;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                if (q.sgn) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, q.ma, q.oa, q.mb, q.ob, iters / 4); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, q.ma, q.oa, q.mb, q.ob, iters / 4);
                hipEventRecord(e0);
                if (q.sgn) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, q.ma, q.oa, q.mb, q.ob, iters); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, q.ma, q.oa, q.mb, q.ob, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            std::printf(" %10.4f", (double)w * 16.0 * iters / (best * 1e-3) / 1e9);
        }
        std::printf("\n");
    }
    return 0;
}

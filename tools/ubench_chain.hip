// Dependent-chain latency of v_mad_u64_u32: one wave per SIMD (and two), NACC independent accumulators per lane.
// A column of a limb product is a chain of up to 9 multiply-adds on ONE accumulator; this measures what interleaving columns buys.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o gpurun_out/ubench_chain ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int NACC>
__global__ void __launch_bounds__(64) chain_k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    uint64_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 0x1234567) out[threadIdx.x] = (uint32_t)s;
}
template <int NACC>
static void run(uint32_t *d, int waves_per_simd, int cus) {
    const int iters = 1 << 16, blocks = cus * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chain_k<NACC>, dim3(blocks), dim3(64), 0, 0, d, 1u, iters / 8);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chain_k<NACC>, dim3(blocks), dim3(64), 0, 0, d, 2u, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_mad = ms * 1e6 / (16.0 * iters);           // per wave
    printf("waves/SIMD %d  accumulators %d  %.3f ms  %.3f ns per multiply-add and wave (%.2f cycles at 2.4 GHz)  chip %.2f T lane-MAC/s\n",
           waves_per_simd, NACC, ms, ns_per_mad, ns_per_mad * 2.4, (double)blocks * 64 * 16.0 * iters / (ms * 1e-3) / 1e12);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    uint32_t *d; hipMalloc(&d, 4096);
    for (int w = 1; w <= 2; ++w) { run<1>(d, w, p.multiProcessorCount); run<2>(d, w, p.multiProcessorCount); run<3>(d, w, p.multiProcessorCount); run<4>(d, w, p.multiProcessorCount); run<8>(d, w, p.multiProcessorCount); }
    // one wave on the whole chip: no power limit in the way
    for (int rep = 0; rep < 1; ++rep) {
        const int iters = 1 << 18;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(chain_k<1>, dim3(1), dim3(64), 0, 0, d, 1u, 1024);
        hipEventRecord(e0, 0); hipLaunchKernelGGL(chain_k<1>, dim3(1), dim3(64), 0, 0, d, 2u, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms1; hipEventElapsedTime(&ms1, e0, e1);
        hipEventRecord(e0, 0); hipLaunchKernelGGL(chain_k<4>, dim3(1), dim3(64), 0, 0, d, 2u, iters); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms4; hipEventElapsedTime(&ms4, e0, e1);
        printf("ONE wave on the chip: 1 accumulator %.3f ns per multiply-add, 4 accumulators %.3f ns\n", ms1 * 1e6 / (16.0 * iters), ms4 * 1e6 / (16.0 * iters));
    }
    return 0;
}

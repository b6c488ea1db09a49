// Instruction-rate microbenchmarks for the integer/fp64 VALU ops a 256-bit Montgomery multiplier can be
// built from on gfx950.  Decides the limb representation (DESIGN.md "leaf arithmetic").
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench && tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 65536;
constexpr int UNROLL = 16;      // instructions per loop body per chain set

// Each kernel: ITERS iterations x UNROLL instructions of one kind. "indep" uses 8 accumulators, "dep" one.
#define KERNEL_BEGIN(name) \
    __global__ void __launch_bounds__(256) name(uint32_t *out, uint32_t seed) { \
        uint64_t t0_ = __builtin_amdgcn_s_memtime(); \
        uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u; \
        uint64_t acc[8]; for (int i = 0; i < 8; ++i) acc[i] = a + i; \
        uint32_t r[8]; for (int i = 0; i < 8; ++i) r[i] = b + i; \
        double d[8]; for (int i = 0; i < 8; ++i) d[i] = 1.0 + i + (a & 7); \
        double da = 1.0000001, db = 0.9999999; \
        for (int it = 0; it < ITERS; ++it) {
#define KERNEL_END \
        } \
        uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i] + r[i] + (uint64_t)d[i]; \
        uint64_t t1_ = __builtin_amdgcn_s_memtime(); \
        if (s == 0x1234567) out[threadIdx.x] = (uint32_t)s; \
        if ((threadIdx.x & 63) == 0) { uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6); if (w < 8192) out[1024 + w] = (uint32_t)(t1_ - t0_); } \
    }

KERNEL_BEGIN(k_mad_u64_u32_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u & 7]) : "v"(a), "v"(b) : "vcc");
KERNEL_END
KERNEL_BEGIN(k_mad_u64_u32_dep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
KERNEL_END
KERNEL_BEGIN(k_mad_u64_u32_dep2)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u & 1]) : "v"(a), "v"(b) : "vcc");
KERNEL_END
KERNEL_BEGIN(k_mad_i64_i32_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[u & 7]) : "v"(a), "v"(b) : "vcc");
KERNEL_END
KERNEL_BEGIN(k_mul_lo_u32_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[u & 7]) : "v"(a));
KERNEL_END
KERNEL_BEGIN(k_mul_hi_u32_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[u & 7]) : "v"(a));
KERNEL_END
KERNEL_BEGIN(k_add_u32_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[u & 7]) : "v"(a));
KERNEL_END
KERNEL_BEGIN(k_add_u32_dep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[0]) : "v"(a));
KERNEL_END
KERNEL_BEGIN(k_add3_u32_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[u & 7]) : "v"(a), "v"(b));
KERNEL_END
KERNEL_BEGIN(k_addc_chain)      // the carry chain of a multi-limb add: strictly serial through vcc
#pragma unroll
    for (int u = 0; u < UNROLL; u += 8) {
        asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n"
                     "v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n"
                     "v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a) : "vcc");
    }
KERNEL_END
KERNEL_BEGIN(k_addc_sgpr_indep)  // carry in/out through arbitrary SGPR pairs (VOP3 form), independent accumulators
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        uint64_t cy;
        asm volatile("v_add_co_u32 %0, %1, %0, %2" : "+v"(r[u & 7]), "=s"(cy) : "v"(a));
    }
KERNEL_END
KERNEL_BEGIN(k_lshl_add_u64_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[u & 7]) : "v"(acc[(u + 1) & 7]));
KERNEL_END
KERNEL_BEGIN(k_lshrrev_b64_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[u & 7]));
KERNEL_END
KERNEL_BEGIN(k_mad_u32_u24_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(r[u & 7]) : "v"(a), "v"(b));
KERNEL_END
KERNEL_BEGIN(k_mul_hi_u32_u24_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r[u & 7]) : "v"(a));
KERNEL_END
KERNEL_BEGIN(k_fma_f64_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[u & 7]) : "v"(da), "v"(db));
KERNEL_END
KERNEL_BEGIN(k_fma_f64_dep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[0]) : "v"(da), "v"(db));
KERNEL_END
KERNEL_BEGIN(k_add_f64_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[u & 7]) : "v"(da));
KERNEL_END
KERNEL_BEGIN(k_cndmask_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[u & 7]) : "v"(a) : );
KERNEL_END
KERNEL_BEGIN(k_cndmask_e64_indep)      // VOP3 form with the condition in an SGPR pair (what `cond ? a : b` compiles to)
    uint64_t m = 0x5555555555555555ull;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[u & 7]) : "v"(a), "s"(m));
KERNEL_END
KERNEL_BEGIN(k_bfi_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[u & 7]) : "v"(a), "v"(b));
KERNEL_END
KERNEL_BEGIN(k_xor_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[u & 7]) : "v"(a));
KERNEL_END
KERNEL_BEGIN(k_mov_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mov_b32 %0, %1" : "=v"(r[u & 7]) : "v"(a));
KERNEL_END
KERNEL_BEGIN(k_mov_dpp_indep)          // partner exchange of the lane-pair mapping
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[u & 7]));
KERNEL_END
KERNEL_BEGIN(k_and_or_indep)           // v_and_or_b32 (VOP3, 3 operands)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[u & 7]) : "v"(a), "v"(b));
KERNEL_END
KERNEL_BEGIN(k_alignbit_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(r[u & 7]) : "v"(a));
KERNEL_END
KERNEL_BEGIN(k_ashr_indep)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(r[u & 7]));
KERNEL_END
KERNEL_BEGIN(k_sub_lit_indep)          // VOP2 with a 32-bit literal
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(r[u & 7]));
KERNEL_END
KERNEL_BEGIN(k_mad_lit_sgpr)           // mad with an SGPR multiplier (the q limbs of the Montgomery reduction)
    uint32_t sq = 0x187cfd47u;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u & 7]) : "v"(a), "s"(sq) : "vcc");
KERNEL_END
// mixed: one quarter-rate mad followed by two full-rate adds on other registers (the CIOS inner step)
KERNEL_BEGIN(k_mix_mad_2add)
#pragma unroll
    for (int u = 0; u < UNROLL; u += 3) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[u & 7]) : "v"(a), "v"(b) : "vcc");
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[u & 7]) : "v"(a));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[(u + 1) & 7]) : "v"(a));
    }
KERNEL_END

typedef void (*kern_t)(uint32_t *, uint32_t);
struct Bench { const char *name; kern_t k; double ops_per_iter; };

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t *out; CK(hipMalloc(&out, (1024 + 8192) * 4));
    std::vector<uint32_t> host(1024 + 8192);

    std::vector<Bench> bs = {
        {"v_mad_u64_u32 indep", k_mad_u64_u32_indep, UNROLL}, {"v_mad_u64_u32 dep1", k_mad_u64_u32_dep, UNROLL},
        {"v_mad_u64_u32 dep2", k_mad_u64_u32_dep2, UNROLL}, {"v_mad_i64_i32 indep", k_mad_i64_i32_indep, UNROLL},
        {"v_mul_lo_u32 indep", k_mul_lo_u32_indep, UNROLL}, {"v_mul_hi_u32 indep", k_mul_hi_u32_indep, UNROLL},
        {"v_add_u32 indep", k_add_u32_indep, UNROLL}, {"v_add_u32 dep", k_add_u32_dep, UNROLL},
        {"v_add3_u32 indep", k_add3_u32_indep, UNROLL}, {"v_add_co/addc chain8", k_addc_chain, UNROLL},
        {"v_add_co sgpr-carry", k_addc_sgpr_indep, UNROLL}, {"v_lshl_add_u64 indep", k_lshl_add_u64_indep, UNROLL},
        {"v_lshrrev_b64 indep", k_lshrrev_b64_indep, UNROLL}, {"v_mad_u32_u24 indep", k_mad_u32_u24_indep, UNROLL},
        {"v_mul_hi_u32_u24 indep", k_mul_hi_u32_u24_indep, UNROLL}, {"v_fma_f64 indep", k_fma_f64_indep, UNROLL},
        {"v_fma_f64 dep", k_fma_f64_dep, UNROLL}, {"v_add_f64 indep", k_add_f64_indep, UNROLL},
        {"v_cndmask vcc", k_cndmask_indep, UNROLL}, {"v_cndmask_e64 sgpr", k_cndmask_e64_indep, UNROLL},
        {"v_bfi_b32", k_bfi_indep, UNROLL}, {"v_xor_b32", k_xor_indep, UNROLL}, {"v_mov_b32", k_mov_indep, UNROLL},
        {"v_mov_b32_dpp quad", k_mov_dpp_indep, UNROLL}, {"v_and_or_b32", k_and_or_indep, UNROLL},
        {"v_alignbit_b32", k_alignbit_indep, UNROLL}, {"v_ashrrev_i32", k_ashr_indep, UNROLL},
        {"v_and_b32 literal", k_sub_lit_indep, UNROLL}, {"v_mad_u64_u32 sgpr", k_mad_lit_sgpr, UNROLL},
        {"mix 1 mad + 2 add", k_mix_mad_2add, 18},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("cyc = s_memtime ticks per wave-instruction as seen by ONE wave; simd = per-SIMD issue interval = cyc / waves-per-SIMD;\n"
           "GHz = effective clock = total wave ticks / wall time\n");
    printf("%-24s", "instruction");
    for (int occ : {1, 2, 4, 8}) printf(" | w/SIMD=%d cyc  simd  GHz", occ);
    printf("\n");
    for (auto &b : bs) {
        printf("%-24s", b.name);
        for (int occ : {1, 2, 4, 8}) {
            int blocks = cus * occ;      // 256 threads = 4 waves = 1 wave per SIMD per block
            b.k<<<blocks, 256>>>(out, 1); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            b.k<<<blocks, 256>>>(out, 2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(host.data(), out, host.size() * 4, hipMemcpyDeviceToHost));
            int nw = blocks * 4 < 8192 ? blocks * 4 : 8192;
            double sum = 0; for (int w = 0; w < nw; ++w) sum += host[1024 + w];
            double ticks = sum / nw;
            double cyc = ticks / ((double)ITERS * b.ops_per_iter);
            printf(" | %13.2f %5.2f %4.2f", cyc, cyc / occ, ticks / (ms * 1e-3) / 1e9);
        }
        printf("\n");
    }
    return 0;
}

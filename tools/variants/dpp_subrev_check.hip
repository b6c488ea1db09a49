// What does v_subrev_u32_dpp compute on gfx950?  (round 5: the bisect of round 4's DPP-fold miscompile ends at ONE instruction,
// `v_subrev_u32_dpp v53, v0, v10 quad_perm:[0,1,0,1]` - profiles/r05_dpp_fold_bisect.txt.)  Each lane has its own x, y; the result of the DPP
// forms of sub / subrev / add is compared on the host with every candidate formula.
//   hipcc --offload-arch=gfx950 -O1 tools/variants/dpp_subrev_check.hip -o /tmp/dpp_subrev_check && /tmp/dpp_subrev_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define DPP01 "quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define DPPX  "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
__global__ void k(const uint32_t *a, const uint32_t *b, uint32_t *o) {
    uint32_t x = a[threadIdx.x], y = b[threadIdx.x], r[6];
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 " DPP01 : "=v"(r[0]) : "v"(x), "v"(y));
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 " DPP01 : "=v"(r[1]) : "v"(x), "v"(y));
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 " DPPX : "=v"(r[2]) : "v"(x), "v"(y));
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 " DPPX : "=v"(r[3]) : "v"(x), "v"(y));
    asm volatile("s_nop 4\n\tv_subrev_co_u32_dpp %0, vcc, %1, %2 " DPP01 : "=v"(r[4]) : "v"(x), "v"(y) : "vcc");
    asm volatile("s_nop 4\n\tv_subrev_u32_e32 %0, %1, %2" : "=v"(r[5]) : "v"(x), "v"(y));
    for (int i = 0; i < 6; ++i) o[threadIdx.x * 6 + i] = r[i];
}
int main() {
    uint32_t ha[64], hb[64], ho[384];
    for (int i = 0; i < 64; ++i) { ha[i] = 1000003u * (i + 1) + 12345u; hb[i] = 7777777u * (i + 3) + 99u; }
    uint32_t *a, *b, *o; hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 1536);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o); hipMemcpy(ho, o, 1536, hipMemcpyDeviceToHost);
    const char *names[6] = {"v_subrev_u32_dpp  d, x, y  [0,1,0,1]", "v_sub_u32_dpp     d, x, y  [0,1,0,1]", "v_subrev_u32_dpp  d, x, y  [1,0,3,2]", "v_sub_u32_dpp     d, x, y  [1,0,3,2]",
                            "v_subrev_co_u32_dpp d, vcc, x, y [0,1,0,1]", "v_subrev_u32_e32  d, x, y  (no DPP)"};
    for (int t = 0; t < 6; ++t) {
        int m[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 64; ++i) {
            const int p = (t == 2 || t == 3) ? (i ^ 1) : ((i & ~3) | (i & 1));            // the lane the DPP operand comes from
            const uint32_t got = ho[i * 6 + t], x = ha[i], y = hb[i], px = ha[p], py = hb[p];
            m[0] += got == y - px;      // S1 - dpp(S0): what the ISA document says subrev is
            m[1] += got == px - y;      // dpp(S0) - S1: what sub is
            m[2] += got == py - x;      // dpp(S1) - S0: the permutation applied to the OTHER operand
            m[3] += got == x - py;
            m[4] += got == y - x;       // no permutation at all
            m[5] += got == x - y;
        }
        std::printf("%-44s lanes equal to:  y - dpp(x) %2d | dpp(x) - y %2d | dpp(y) - x %2d | x - dpp(y) %2d | y - x %2d | x - y %2d\n", names[t], m[0], m[1], m[2], m[3], m[4], m[5]);
    }
    return 0;
}

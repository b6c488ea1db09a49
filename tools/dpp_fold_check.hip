#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t *a, const uint32_t *b, uint32_t *o) {
    uint32_t x = a[threadIdx.x], y = b[threadIdx.x];
    uint32_t m = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x44, 0xF, 0xF, true);      // quad_perm [0,1,0,1]
    uint32_t ref_sub = m - y, ref_rev = y - m;
    uint32_t f_sub, f_rev;
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(f_sub) : "v"(x), "v"(y));
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(f_rev) : "v"(x), "v"(y));
    // the producer right before (VOP3 v_alignbit_b32), two DPP moves between - the shape in the kernel
    uint32_t hi = x >> 7, lo = y * 2654435761u, t, f2, d1, d2;
    asm volatile("v_alignbit_b32 %0, %4, %5, 29\n\tv_mov_b32_dpp %2, %5 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_mov_b32_dpp %3, %4 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "v_sub_u32_dpp %1, %0, %6 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0"
                 : "=&v"(t), "=&v"(f2), "=&v"(d1), "=&v"(d2) : "v"(hi), "v"(lo), "v"(y));
    uint32_t tt = (uint32_t)((((uint64_t)hi << 32) | lo) >> 29);
    uint32_t r2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)tt, 0x44, 0xF, 0xF, true) - y;
    o[threadIdx.x * 8 + 0] = ref_sub; o[threadIdx.x * 8 + 1] = f_sub; o[threadIdx.x * 8 + 2] = ref_rev; o[threadIdx.x * 8 + 3] = f_rev;
    o[threadIdx.x * 8 + 4] = r2; o[threadIdx.x * 8 + 5] = f2; o[threadIdx.x * 8 + 6] = t; o[threadIdx.x * 8 + 7] = tt;
}
int main() {
    uint32_t ha[64], hb[64], ho[512];
    for (int i = 0; i < 64; ++i) { ha[i] = 1000003u * (i + 1) + 12345u; hb[i] = 7777777u * (i + 3); }
    uint32_t *a, *b, *o; hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 2048);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o); hipMemcpy(ho, o, 2048, hipMemcpyDeviceToHost);
    int bad_sub = 0, bad_rev = 0, bad2 = 0, swapped_sub = 0;
    for (int i = 0; i < 64; ++i) { bad_sub += ho[8*i] != ho[8*i+1]; bad_rev += ho[8*i+2] != ho[8*i+3]; bad2 += ho[8*i+4] != ho[8*i+5]; swapped_sub += ho[8*i+1] == ho[8*i+2]; }
    std::printf("v_sub_u32_dpp   != mov_dpp ; sub      in %d of 64 lanes (equals the REVERSED difference in %d)\n", bad_sub, swapped_sub);
    std::printf("v_subrev_u32_dpp != mov_dpp ; subrev  in %d of 64 lanes\n", bad_rev);
    std::printf("alignbit ; 2 dpp moves ; v_sub_u32_dpp != reference in %d of 64 lanes\n", bad2);
    for (int i = 0; i < 4; ++i) std::printf("lane %d: ref_sub %08x folded %08x | ref_rev %08x folded %08x | shape ref %08x folded %08x\n", i, ho[8*i], ho[8*i+1], ho[8*i+2], ho[8*i+3], ho[8*i+4], ho[8*i+5]);
    return 0;
}

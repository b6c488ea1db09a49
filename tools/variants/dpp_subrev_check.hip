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
// Round 6: is it the ASSEMBLER that places the DPP operand, or the silicon?  The same instructions HAND-ENCODED (.long) per the gfx950 VOP2 + DPP16
// layout, with fixed registers v40 = x, v41 = y, result in v42:
//   dword 0 (VOP2):  [8:0] src0 = 0xFA (= "a DPP dword follows")   [16:9] vsrc1   [24:17] vdst   [30:25] opcode (v_sub_u32 0x35, v_subrev_u32 0x36)   [31] 0
//   dword 1 (DPP16): [7:0] the VGPR that IS src0 - the only operand the encoding can attach a lane permutation to -   [16:8] dpp_ctrl (quad_perm [0,1,0,1] = 0x44)
//                    [19] bound_ctrl   [27:24] bank_mask   [31:28] row_mask
// llvm-mc -mcpu=gfx950 -show-encoding gives exactly these bytes for `v_subrev_u32_dpp v42, v40, v41 quad_perm:[0,1,0,1] ...` (first variant below), so the
// assembler puts the DPP register where the ISA says src0 goes.  Variant 2 swaps the two registers in the ENCODING (src0 = v41 = y with the permutation, vsrc1 = x).
#define VOP2_DPP(op, vdst, vsrc1) ((op) << 25 | (vdst) << 17 | (vsrc1) << 9 | 0xFA)
#define DPP16(src0, ctrl) ((src0) | (ctrl) << 8 | 1u << 19 | 0xFu << 24 | 0xFu << 28)
#define STR2(x) #x
#define STR(x) STR2(x)
#define RUN_ENC(dst, w0, w1) asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, 0\n\ts_nop 4\n\t.long " STR(w0) "\n\t.long " STR(w1) "\n\ts_nop 4\n\tv_mov_b32 %0, v42" \
                                          : "=v"(dst) : "v"(x), "v"(y) : "v40", "v41", "v42")
__global__ void k_enc(const uint32_t *a, const uint32_t *b, uint32_t *o) {
    uint32_t x = a[threadIdx.x], y = b[threadIdx.x], r[4];
    RUN_ENC(r[0], 0x6C5452FA, 0xFF084428);      // v_subrev_u32  vdst v42, src0 = DPP(v40 = x), vsrc1 = v41 = y      [0,1,0,1]
    RUN_ENC(r[1], 0x6C5450FA, 0xFF084429);      // v_subrev_u32  vdst v42, src0 = DPP(v41 = y), vsrc1 = v40 = x
    RUN_ENC(r[2], 0x6A5452FA, 0xFF084428);      // v_sub_u32     vdst v42, src0 = DPP(v40 = x), vsrc1 = v41 = y
    RUN_ENC(r[3], 0x6A5450FA, 0xFF084429);      // v_sub_u32     vdst v42, src0 = DPP(v41 = y), vsrc1 = v40 = x
    for (int i = 0; i < 4; ++i) o[threadIdx.x * 4 + i] = r[i];
}
static_assert(VOP2_DPP(0x36u, 42u, 41u) == 0x6C5452FAu && DPP16(40u, 0x44u) == 0xFF084428u, "hand encoding of v_subrev_u32_dpp v42, v40, v41");
static_assert(VOP2_DPP(0x36u, 42u, 40u) == 0x6C5450FAu && DPP16(41u, 0x44u) == 0xFF084429u, "hand encoding of v_subrev_u32_dpp v42, v41, v40");
static_assert(VOP2_DPP(0x35u, 42u, 41u) == 0x6A5452FAu && VOP2_DPP(0x35u, 42u, 40u) == 0x6A5450FAu, "hand encoding of v_sub_u32_dpp");
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
    // ---- hand-encoded instructions
    uint32_t *o2; hipMalloc(&o2, 1024); uint32_t h2[256];
    hipLaunchKernelGGL(k_enc, dim3(1), dim3(64), 0, 0, a, b, o2); hipMemcpy(h2, o2, 1024, hipMemcpyDeviceToHost);
    const char *n2[4] = {".long subrev: src0 = DPP(x), vsrc1 = y", ".long subrev: src0 = DPP(y), vsrc1 = x", ".long sub:    src0 = DPP(x), vsrc1 = y", ".long sub:    src0 = DPP(y), vsrc1 = x"};
    std::printf("hand-encoded VOP2 + DPP16 words (quad_perm [0,1,0,1]); S0 = the register in the DPP dword, S1 = vsrc1:\n");
    for (int t = 0; t < 4; ++t) {
        int m[4] = {0, 0, 0, 0};
        for (int i = 0; i < 64; ++i) {
            const int p = (i & ~3) | (i & 1);
            const bool swapped = (t & 1) != 0;
            const uint32_t got = h2[i * 4 + t];
            const uint32_t s0 = swapped ? hb[i] : ha[i], s1 = swapped ? ha[i] : hb[i], ps0 = swapped ? hb[p] : ha[p], ps1 = swapped ? ha[p] : hb[p];
            m[0] += got == s1 - ps0;    // S1 - dpp(S0): subrev as documented
            m[1] += got == ps1 - s0;    // dpp(S1) - S0: subrev with the permutation on the OTHER source
            m[2] += got == ps0 - s1;    // dpp(S0) - S1: sub as documented
            m[3] += got == s0 - ps1;    // S0 - dpp(S1)
        }
        std::printf("%-44s lanes equal to:  S1 - dpp(S0) %2d | dpp(S1) - S0 %2d | dpp(S0) - S1 %2d | S0 - dpp(S1) %2d\n", n2[t], m[0], m[1], m[2], m[3]);
    }
    return 0;
}

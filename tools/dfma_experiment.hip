// Experiment (round-3 verdict item 1, SURVEY section 7 "which multiplier"): the Fq Montgomery product on the FP64 FMA unit - five
// 52-bit limbs held as doubles, every 52 x 52-bit partial product split into its high and low halves by two v_fma_f64 with the
// wave's f64 rounding mode set toward zero (Emmart / Zheng / Weems, "Faster modular exponentiation using double precision floating
// point arithmetic on the GPU", ARITH 2018) - against the shipped leaf (fe.hpp / fe_asm.hpp: nine 29-bit limbs, one v_mad_u64_u32
// per partial product, a 64-bit column accumulator that never needs a carry instruction).
//
//   hi' = fma_rz(a, b, 2^104)                 = 2^104 + floor(a b / 2^52) 2^52          (ulp of [2^104, 2^105) is 2^52)
//   lo' = fma_rz(a, b, (2^104 + 2^52) - hi')  = 2^52  + (a b mod 2^52)                  (exact: the addend is a multiple of 2^52)
// The mantissa fields of hi' and lo' ARE the two halves of the product; column sums are formed by adding the bit patterns as 64-bit
// integers (the exponent fields add up to a constant per column, preloaded negated into the accumulator).  Montgomery radix 2^260,
// product scanning with interleaved reduction like fe_mul: per limb product 2 v_fma_f64 + 1 v_add_f64 + 2 64-bit integer adds, where
// the integer leaf spends ONE v_mad_u64_u32.  50 limb products (a b and m q) against 162: that is the whole question.
//
// What is measured: (1) bit identity at the boundary - both leaves run the same chain of dependent Fq2 products in the lane-pair
// mapping (fq2.hpp: each lane evaluates own_a u + partner_a v with one reduction) on 2^20 random + edge operands (0, 1, q - 1, and
// the unreduced values < 2q the chain itself feeds back) and must store the same canonical bytes of the reference's radix-2^256
// image (src/arith.rs:257-263,481-503: canonical at the boundary); (2) time per dependent dual product at 1, 2 and 4 waves per SIMD
// on all 1024 SIMDs; (3) the instruction mix of both loops (tools/isa_mix.py on this binary).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bn_amd/csrc tools/dfma_experiment.hip -o build_variants/dfma_experiment
#define BN_INLINE_ALL 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "fe.hpp"

using namespace bn254;

// ---------------------------------------------------------------------------------------------------------------- DFMA leaf
struct D5 { double l[5]; };                    // value = sum l[i] 2^(52 i), every l[i] an integer in [0, 2^52)
struct DfmaConsts {                            // kernel argument (host-computed from q, checked against the 29-bit constants)
    double q[5];                               // q in 52-bit limbs
    double qinv;                               // -q^-1 mod 2^52
    double k_in[5];                            // 2^264 mod q: mont(X, k_in) = X 2^4   (X = x 2^256, the reference's image)
    double k_out[5];                           // 2^256 mod q: mont(v, k_out) = v 2^-4
};

constexpr uint64_t EXP104 = (uint64_t)(1023 + 104) << 52, EXP52 = (uint64_t)(1023 + 52) << 52, M52 = (1ull << 52) - 1;

__device__ __forceinline__ double u2d(uint64_t x) { return __longlong_as_double((long long)x); }
__device__ __forceinline__ uint64_t d2u(double x) { return (uint64_t)__double_as_longlong(x); }
// the three FP64 instructions of a limb product, as asm so that nothing is folded, contracted or re-associated under the
// non-default rounding mode
__device__ __forceinline__ double fma_rz(double a, double b, double c) { double r; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ double fma_rz_k(double a, double b, double c) { double r; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c)); return r; }
__device__ __forceinline__ double sub_k(double c, double x) { double r; asm("v_add_f64 %0, %1, -%2" : "=v"(r) : "s"(c), "v"(x)); return r; }
__device__ __forceinline__ double add_k(double c, double x) { double r; asm("v_add_f64 %0, %1, %2" : "=v"(r) : "s"(c), "v"(x)); return r; }
__device__ __forceinline__ void add64(uint64_t &acc, double x) { asm("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(acc) : "v"(x)); }
// integer in [0, 2^52) -> double (exact): splice it under the exponent of 2^52, subtract 2^52
__device__ __forceinline__ double int52_to_double(uint64_t x) { return add_k(-4503599627370496.0, u2d(EXP52 | (x & M52))); }

// one limb product into the column accumulators: lo half -> col, hi half -> col_up.  ONE asm statement for its five instructions (the
// compiler answers every asm statement with an s_nop - it cannot see which hazards the statement leaves open)
__device__ __forceinline__ void term(uint64_t &col, uint64_t &col_up, double a, double b) {
    const double c1 = u2d(EXP104), c2 = u2d(EXP104) + 4503599627370496.0;          // 2^104, 2^104 + 2^52 (compile-time constants)
    double hi, lo;
    asm("v_fma_f64 %2, %4, %5, %6\n\tv_add_f64 %3, %7, -%2\n\tv_fma_f64 %3, %4, %5, %3\n\tv_lshl_add_u64 %1, %2, 0, %1\n\tv_lshl_add_u64 %0, %3, 0, %0"
        : "+v"(col), "+v"(col_up), "=&v"(hi), "=&v"(lo) : "v"(a), "v"(b), "s"(c1), "s"(c2));
}
// (a u + c v) / 2^260 mod q, one reduction (DUAL = false: a u / 2^260).  Result limbs < 2^52, value < q + (sum of the products) / 2^260.
template <bool DUAL>
__device__ __forceinline__ D5 dfma_mul2(const D5 &a, const D5 &u, const D5 &c, const D5 &v, const DfmaConsts &K) {
    // column k receives n_lo(k) low halves and n_hi(k) high halves; their exponent fields are taken out up front
    uint64_t col[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        // products a_i u_j: i + j = k gives a low half to column k, i + j = k - 1 a high half; the same counts again for m_i q_j
        const int nlo = (k <= 4 ? k + 1 : (k <= 8 ? 9 - k : 0)), nhi = (k >= 1 && k <= 5 ? k : (k >= 6 && k <= 9 ? 10 - k : 0));
        const int f = DUAL ? 3 : 2;                                                 // a u (+ c v) + m q
        col[k] = 0 - ((uint64_t)(f * nlo) * EXP52 + (uint64_t)(f * nhi) * EXP104);
    }
    double m[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            term(col[k], col[k + 1], a.l[i], u.l[k - i]);
            if (DUAL) term(col[k], col[k + 1], c.l[i], v.l[k - i]);
        }
#pragma unroll
        for (int i = 0; i < k; ++i) term(col[k], col[k + 1], m[i], K.q[k - i]);
        // m_k = col_k * (-q^-1) mod 2^52: the low half of one more limb product (col[k] here is exact: every low term of this column
        // except m_k q_0 has arrived, and so has every high term)
        {
            // the exponent fields still owed to this column by the m_k q_0 term are added back for the moment
            const uint64_t have = col[k] + EXP52;
            const double cd = int52_to_double(have);
            const double c1 = u2d(EXP104), c2 = u2d(EXP104) + 4503599627370496.0;
            const double hi = fma_rz_k(cd, K.qinv, c1);
            const double lo = fma_rz(cd, K.qinv, sub_k(c2, hi));
            m[k] = add_k(-4503599627370496.0, lo);                                 // lo - 2^52: the integer m_k as a double
        }
        term(col[k], col[k + 1], m[k], K.q[0]);
        col[k + 1] += col[k] >> 52;                                                 // the low 52 bits are zero now
    }
#pragma unroll
    for (int k = 5; k < 9; ++k) {
#pragma unroll
        for (int i = k - 4; i <= 4; ++i) {
            term(col[k], col[k + 1], a.l[i], u.l[k - i]);
            if (DUAL) term(col[k], col[k + 1], c.l[i], v.l[k - i]);
            term(col[k], col[k + 1], m[i], K.q[k - i]);
        }
        col[k + 1] += col[k] >> 52;
    }
    D5 r;
#pragma unroll
    for (int k = 5; k < 9; ++k) r.l[k - 5] = int52_to_double(col[k]);
    r.l[4] = int52_to_double(col[9]);                                               // the value is below 2^256: the top limb fits
    return r;
}

// reference image (8 x u32, canonical) <-> five 52-bit limbs
__device__ __forceinline__ D5 d5_unpack(const uint32_t *w) {
    uint64_t x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    D5 r;
    r.l[0] = int52_to_double(x[0]);
    r.l[1] = int52_to_double((x[0] >> 52) | (x[1] << 12));
    r.l[2] = int52_to_double((x[1] >> 40) | (x[2] << 24));
    r.l[3] = int52_to_double((x[2] >> 28) | (x[3] << 36));
    r.l[4] = int52_to_double(x[3] >> 16);
    return r;
}
__device__ __forceinline__ void d5_store_canonical(const D5 &a, uint32_t *w, const DfmaConsts &K) {
    uint64_t l[5], ql[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { l[i] = (uint64_t)a.l[i]; ql[i] = (uint64_t)K.q[i]; }
    // value < 2q: one conditional subtraction
    uint64_t d[5]; int64_t br = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) { int64_t s = (int64_t)l[i] - (int64_t)ql[i] + br; d[i] = (uint64_t)s & M52; br = s >> 52; }
    if (br == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) l[i] = d[i];
    }
    uint64_t x[4] = {l[0] | (l[1] << 52), (l[1] >> 12) | (l[2] << 40), (l[2] >> 24) | (l[3] << 28), (l[3] >> 36) | (l[4] << 16)};
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[2 * i] = (uint32_t)x[i]; w[2 * i + 1] = (uint32_t)(x[i] >> 32); }
}
__device__ __forceinline__ D5 d5_partner(const D5 &x) {                             // the other lane of the lane pair (DPP quad_perm [1,0,3,2])
    D5 r;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const uint64_t v = d2u(x.l[i]);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, 0xB1, 0xF, 0xF, true);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), 0xB1, 0xF, 0xF, true);
        r.l[i] = u2d((uint64_t)lo | ((uint64_t)hi << 32));
    }
    return r;
}
__device__ __forceinline__ Fe fe_partner(const Fe &x) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[i], 0xB1, 0xF, 0xF, true);
    return r;
}
__device__ __forceinline__ void set_f64_round_toward_zero() {
    // MODE register (hwreg 1), bits [3:2] = rounding of f64 / f16: 3 = toward zero.  "memory": the loads that feed the FP
    // instructions are issued after it
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------- the chains
// element e = one Fq2 pair (a, b) on lanes 2e (c0) and 2e + 1 (c1); `rounds` times a <- a * b (the dependent chain of a Miller
// loop or exponentiation step), then the canonical image of a.  Inputs: 16 u32 per Fq2 (c0, c1 as the reference's [u64; 4]).
// each lane: own_a * u + partner_a * v with (u, v) = (b0, -b1) on the even lane and (b1, b0) on the odd lane
__global__ void __launch_bounds__(64) chain_int29(const uint32_t *a_in, const uint32_t *b_in, uint32_t *out, uint32_t n, int rounds) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    uint32_t e = t >> 1;
    const bool odd = t & 1, live = e < n;
    if (!live) e = n - 1;
    Fe a = fe_from_u32x8(a_in + 16u * e + (odd ? 8u : 0u));
    const Fe b_own = fe_from_u32x8(b_in + 16u * e + (odd ? 8u : 0u));
    const Fe b_par = fe_partner(b_own);
    const Fe u = fe_select(odd, b_own, b_par);                                      // even: b0, odd: b0
    const Fe v = fe_select(odd, fe_norm(fe_neg<1, 3>(b_par)), b_own);               // even: -b1, odd: b1 ... see below
    // even lane: a0 b0 + a1 (-b1);  odd lane (own = a1, partner = a0): a1 b0 + a0 b1
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) a = fe_mul2(a, u, fe_partner(a), v);
    if (live) fe_to_u32x8(a, out + 16u * e + (odd ? 8u : 0u));
}
__global__ void __launch_bounds__(64) chain_dfma(const uint32_t *a_in, const uint32_t *b_in, uint32_t *out, uint32_t n, int rounds, DfmaConsts K) {
    set_f64_round_toward_zero();
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    uint32_t e = t >> 1;
    const bool odd = t & 1, live = e < n;
    if (!live) e = n - 1;
    D5 kin, kout;
#pragma unroll
    for (int i = 0; i < 5; ++i) { kin.l[i] = K.k_in[i]; kout.l[i] = K.k_out[i]; }
    const D5 z = {{0, 0, 0, 0, 0}};
    D5 a = dfma_mul2<false>(d5_unpack(a_in + 16u * e + (odd ? 8u : 0u)), kin, z, z, K);
    const D5 b_own = dfma_mul2<false>(d5_unpack(b_in + 16u * e + (odd ? 8u : 0u)), kin, z, z, K);
    const D5 b_par = d5_partner(b_own);
    // -b1 as 2q - b1, limb-wise with a borrow chain (b1 < 2q; integers in doubles up to 2^53 are exact)
    D5 nb;
    {
        int64_t br = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int64_t s = 2 * (int64_t)K.q[i] - (int64_t)b_par.l[i] + br;
            if (i < 4) { nb.l[i] = (double)(s & (int64_t)M52); br = s >> 52; } else { nb.l[i] = (double)s; }
        }
    }
    D5 u, v;
#pragma unroll
    for (int i = 0; i < 5; ++i) { u.l[i] = odd ? b_par.l[i] : b_own.l[i]; v.l[i] = odd ? b_own.l[i] : nb.l[i]; }
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) a = dfma_mul2<true>(a, u, d5_partner(a), v, K);
    if (live) d5_store_canonical(dfma_mul2<false>(a, kout, z, z, K), out + 16u * e + (odd ? 8u : 0u), K);
}

// ---------------------------------------------------------------------------------------------------------------- host
typedef unsigned __int128 u128;
struct U256 { uint64_t w[4]; };
static const U256 Q = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
static bool geq(const U256 &a, const U256 &b) { for (int i = 3; i >= 0; --i) { if (a.w[i] != b.w[i]) return a.w[i] > b.w[i]; } return true; }
static U256 sub(const U256 &a, const U256 &b) { U256 r; u128 br = 0; for (int i = 0; i < 4; ++i) { u128 t = (u128)a.w[i] - b.w[i] - br; r.w[i] = (uint64_t)t; br = (t >> 64) & 1; } return r; }
static U256 dbl_mod(const U256 &a) { U256 r; uint64_t c = 0; for (int i = 0; i < 4; ++i) { r.w[i] = (a.w[i] << 1) | c; c = a.w[i] >> 63; } if (c || geq(r, Q)) r = sub(r, Q); return r; }
static U256 pow2_mod(int k) { U256 r = {{1, 0, 0, 0}}; for (int i = 0; i < k; ++i) r = dbl_mod(r); return r; }
static void limbs52(const U256 &x, double *l) {
    l[0] = (double)(x.w[0] & ((1ull << 52) - 1));
    l[1] = (double)(((x.w[0] >> 52) | (x.w[1] << 12)) & ((1ull << 52) - 1));
    l[2] = (double)(((x.w[1] >> 40) | (x.w[2] << 24)) & ((1ull << 52) - 1));
    l[3] = (double)(((x.w[2] >> 28) | (x.w[3] << 36)) & ((1ull << 52) - 1));
    l[4] = (double)(x.w[3] >> 16);
}
static uint64_t sm(uint64_t &s) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main(int argc, char **argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    DfmaConsts K;
    limbs52(Q, K.q);
    { uint64_t q0 = Q.w[0], x = 1; for (int i = 0; i < 6; ++i) x *= 2 - q0 * x; K.qinv = (double)((0 - x) & ((1ull << 52) - 1)); }   // Newton: q0 x = 1 mod 2^64
    limbs52(pow2_mod(264), K.k_in);
    limbs52(pow2_mod(256), K.k_out);

    // ---- (1) bit identity: 2^20 Fq2 pairs; the first 81 are every combination of {0, 1, q - 1} in (a0, a1, b0, b1), in Montgomery
    // image terms: the raw words 0, 1, q - 1 (any canonical word string is a valid image)
    const uint32_t n = quick ? 1u << 14 : 1u << 20;
    std::vector<uint64_t> ha((size_t)n * 8), hb((size_t)n * 8);
    uint64_t seed = 2024;
    for (size_t i = 0; i < ha.size(); ++i) { ha[i] = sm(seed); hb[i] = sm(seed); if (i % 4 == 3) { ha[i] %= 0x30644e72e131a029ull; hb[i] %= 0x30644e72e131a029ull; } }
    const U256 edge[3] = {{{0, 0, 0, 0}}, {{1, 0, 0, 0}}, sub(Q, U256{{1, 0, 0, 0}})};
    for (int c = 0; c < 81; ++c) {
        const int i0 = c % 3, i1 = (c / 3) % 3, i2 = (c / 9) % 3, i3 = c / 27;
        memcpy(&ha[(size_t)c * 8], edge[i0].w, 32); memcpy(&ha[(size_t)c * 8 + 4], edge[i1].w, 32);
        memcpy(&hb[(size_t)c * 8], edge[i2].w, 32); memcpy(&hb[(size_t)c * 8 + 4], edge[i3].w, 32);
    }
    uint32_t *da, *db, *o1, *o2;
    const size_t bytes = (size_t)n * 64;
    hipMalloc(&da, bytes); hipMalloc(&db, bytes); hipMalloc(&o1, bytes); hipMalloc(&o2, bytes);
    hipMemcpy(da, ha.data(), bytes, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), bytes, hipMemcpyHostToDevice);
    bool all_equal = true;
    for (int rounds : {1, 2, 7}) {
        hipMemset(o1, 0xAA, bytes); hipMemset(o2, 0x55, bytes);
        hipLaunchKernelGGL(chain_int29, dim3(n * 2 / 64), dim3(64), 0, 0, da, db, o1, n, rounds);
        hipLaunchKernelGGL(chain_dfma, dim3(n * 2 / 64), dim3(64), 0, 0, da, db, o2, n, rounds, K);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        std::vector<uint64_t> r1((size_t)n * 8), r2((size_t)n * 8);
        hipMemcpy(r1.data(), o1, bytes, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, bytes, hipMemcpyDeviceToHost);
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < (size_t)n; ++i) if (memcmp(&r1[i * 8], &r2[i * 8], 64)) { if (!bad) first = i; ++bad; }
        // sanity of the check itself: results are canonical (< q) and not all zero
        bool canon = true, nonzero = false;
        for (size_t i = 0; i < (size_t)n * 2; ++i) { U256 x; memcpy(x.w, &r1[i * 4], 32); if (geq(x, Q)) canon = false; if (x.w[0] | x.w[1] | x.w[2] | x.w[3]) nonzero = true; }
        printf("bit identity, %u Fq2 elements (81 edge combinations of 0, 1, q-1 first), %d dependent product(s): %s", n, rounds, bad ? "DIFFERENT" : "equal");
        if (bad) printf(" (%zu elements, first %zu)", bad, first);
        printf("%s\n", canon && nonzero ? "" : "  [check broken: non-canonical or all-zero reference output]");
        all_equal = all_equal && !bad && canon && nonzero;
    }
    // ---- (2) time per dependent dual product, W waves per SIMD on 1024 SIMDs
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int simds = prop.multiProcessorCount * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = quick ? 2000 : 20000;
    printf("\n%d CUs; %d dependent dual products (own_a u + partner_a v, one reduction) per lane; best of 3 launches\n", prop.multiProcessorCount, rounds);
    printf("%12s %14s %14s %9s %16s %16s\n", "waves/SIMD", "int29 ns/prod", "dfma ns/prod", "ratio", "int29 Gprod/s", "dfma Gprod/s");
    for (int wps : {1, 2, 4}) {
        const uint32_t waves = (uint32_t)simds * wps, ne = waves * 32;
        if (ne > n) { printf("%12d skipped (needs %u elements)\n", wps, ne); continue; }
        float best[2] = {1e30f, 1e30f};
        for (int rep = 0; rep < 4; ++rep) {
            float ms;
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(chain_int29, dim3(waves), dim3(64), 0, 0, da, db, o1, ne, rounds);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best[0]) best[0] = ms;
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(chain_dfma, dim3(waves), dim3(64), 0, 0, da, db, o2, ne, rounds, K);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best[1]) best[1] = ms;
        }
        const double lanes = (double)waves * 64;
        printf("%12d %14.1f %14.1f %9.2f %16.2f %16.2f\n", wps, best[0] * 1e6 / rounds, best[1] * 1e6 / rounds, best[0] / best[1],
               lanes * rounds / (best[0] * 1e-3) * 1e-9, lanes * rounds / (best[1] * 1e-3) * 1e-9);
    }
    printf("ratio > 1: the DFMA leaf is faster.  Gprod/s counts lane-level dual products (each = 1.5 Fq-product equivalents).\n");
    printf("%s\n", all_equal ? "all results bit-identical between the two leaves" : "MISMATCH");
    return all_equal ? 0 : 1;
}

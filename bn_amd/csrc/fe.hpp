// Base-field arithmetic of the HIP pairing engine: Fq elements as 9 x 29-bit limbs ("Fe"), Montgomery radix 2^261.
//
// Replaces, for the GPU, src/arith.rs:183-503 + src/fields/fp.rs:9-182 of the reference (U256 / Fq with 4 x u64 limbs,
// radix 2^256, every value canonical).  Why a different number system (measured, profiles/r01_ubench_valu_rates.txt):
//   * v_mad_u64_u32 is the best multiplier gfx950 has (0.44-0.46 G wave-instr/s/SIMD), and it adds a 64-bit addend for
//     free, but every carry-writing add (v_add_co/v_addc) runs at half the plain-add rate.  With 29-bit limbs a whole
//     column of 18 partial products accumulates in one 64-bit register with NO carry handling at all.
//   * 9 x 29 = 261 bits leaves q/2^261 ~ 1/169 of head-room, so sums, differences and even products of unreduced values
//     stay representable: additions are 9 plain v_add_u32, reduction mod q happens only where the bounds demand it.
//
// Representation invariants (tracked and ENFORCED at run time in the host simulation build, -DBN_BOUNDS; the control flow
// of the engine is data independent, so one simulated run exercises every bound):
//   limb bound  lb : l[i] <= lb * (2^29 - 1)  for i < 8      (lb = 1: "normalized")
//   value bound vb : value <= vb * q                           (the top limb l[8] holds everything above 2^232)
// Montgomery products return (lb, vb) = (1, 2).  Values are only made canonical (< q) when they leave the engine
// (fe_to_u32x8), where they are converted back to the reference's radix-2^256 image, so the bytes at the C ABI are
// exactly the reference's.
#pragma once
#include <stdint.h>

// BN_FN     small glue, always inlined
// BN_LEAF   the multiplier-sized leaves (fe_mul, fe_mul2, fe_lc3, fe_inverse): real functions by default (operands travel in
//           VGPRs: an Fe is 9 dwords, under the 16-dword by-value limit of the AMDGPU calling convention), so the hot code
//           of a whole pairing stays inside the instruction cache instead of being replicated 19 000 times
// BN_COARSE Fq6/Fq12-sized steps: real functions by default; their operands live in private (scratch) memory in the one-lane
//           mapping and in VGPRs in the lane-pair mapping
// A translation unit that defines BN_INLINE_ALL before including this header (every kernel file of the lane-pair, four-lane, wave and
// scalar-multiplication kernels) gets all three classes force-inlined: without calls there is no argument marshalling (v_mov was ~45 % of
// the non-leaf instructions) and no caller-/callee-saved split of the register file (+7 %, profiles/r01g_*).
#if defined(BN_HOSTSIM)
#define BN_FN inline
#define BN_LEAF inline
#define BN_LEAF_MUL inline
#define BN_LEAF_RED inline
#define BN_COARSE inline
#define BN_OUTER inline
#define BN254_CONSTANT constexpr
#define BN_COMPILER_FENCE() ((void)0)
#else
#include <hip/hip_runtime.h>
#define BN_FN __device__ __forceinline__
#define BN_LEAF __device__ __noinline__ inline
#ifdef BN_INLINE_ALL
#define BN_LEAF_MUL __device__ __forceinline__      // the multiplier-sized leaves (fe_mul, f2b_mul, f2b_sqr)
#define BN_LEAF_RED __device__ __forceinline__      // the reduction-sized leaves (fe_lc3 family)
#define BN_COARSE __device__ __forceinline__
#else
#define BN_LEAF_MUL BN_LEAF
#define BN_LEAF_RED BN_LEAF
#define BN_COARSE __device__ __noinline__ inline
#endif
// BN_OUTER  big, rarely executed steps (inversions, Frobenius maps, the straight-line part of the final exponentiation):
//           always real functions, so the hot loops are not diluted by their code
#define BN_OUTER __device__ __noinline__ inline
#define BN254_CONSTANT __device__ constexpr
// keeps loads that follow in the source from being scheduled above this point (used where early loads only cause spills)
#define BN_COMPILER_FENCE() asm volatile("" ::: "memory")
#endif
// Hooks of the two-waves-per-SIMD priority hand-over (bn254_kernels_b.hip defines both before including this header): executed at the top
// of every step of the Miller loop resp. of the three exponentiation loops of the final exponentiation, with the step number and the count
#ifndef BN_MILLER_HOOK
#define BN_MILLER_HOOK(step, total) ((void)0)
#endif
#ifndef BN_EXP_HOOK
#define BN_EXP_HOOK(step, total) ((void)0)
#endif
#ifndef BN_MUL_HOOK
#define BN_MUL_HOOK(step, total) ((void)0)
#endif
#ifndef BN_G1_HOOK
#define BN_G1_HOOK(step, total) ((void)0)
#endif
#include "bn254_constants.hpp"
// GPU builds run the multiplier leaves and the 64-bit chains of the fused reductions as inline-asm instruction chains (fe_asm.hpp: same
// arithmetic, fixed instruction order); the host simulation runs the C++ bodies, which also carry the bound checks

#if defined(BN_BOUNDS)
#include <cstdio>
#include <cstdlib>
#define BN_REQUIRE(cond, what)                                                                              \
    do {                                                                                                    \
        if (!(cond)) {                                                                                      \
            std::fprintf(stderr, "BN_BOUNDS violation: %s  [%s]  at %s:%d\n", what, #cond, __FILE__, __LINE__); \
            std::abort();                                                                                   \
        }                                                                                                   \
    } while (0)
#define BN_SETB(x, LBV, VBV) ((x).lb = (LBV), (x).vb = (VBV), (x).sg = false)
#define BN_IFB(...) __VA_ARGS__
// `macs`: the multiply instructions the GPU leaf issues for the operation (v_mad_u64_u32 / v_mad_i64_i32 / v_mul_lo / v_mul_hi): the EXECUTED
// multiply-adds behind bench.py's roofline.frac_executed (profiles/executed_chain_lengths.json "mac_instructions_per_unit")
namespace bn254 { struct OpCounts { unsigned long mul, mul2, lc3, lc3w, norm, addsub, reduce, select, macs; }; inline OpCounts &op_counts() { static OpCounts c{}; return c; } }
#define BN_COUNT(f) (++bn254::op_counts().f)
#define BN_COUNT_MACS(n) (bn254::op_counts().macs += (unsigned long)(n))
#else
#define BN_REQUIRE(cond, what) ((void)0)
#define BN_SETB(x, LBV, VBV) ((void)0)
#define BN_IFB(...)
#define BN_COUNT(f) ((void)0)
#define BN_COUNT_MACS(n) ((void)0)
#endif

namespace bn254 {

constexpr uint32_t MASK29 = 0x1fffffffu;

struct Fe {
    uint32_t l[9];
#if defined(BN_BOUNDS)
    uint32_t lb = 1, vb = 1;
    bool sg = false;      // "signed lazy": limbs are int32 in two's complement, the value may be negative (only fe_lc3 takes these)
#endif
};

#if defined(BN_BOUNDS)
// Host-simulation soundness check: the bounds an operation CLAIMS for its result (lb, vb, sg) are compared with the limbs it
// actually produced, so an error in the bound analysis (not just in the arithmetic) aborts the simulated run.
inline void bn_verify_actual(const Fe &f, const char *where) {
    // limbs
    for (int i = 0; i < 8; ++i) {
        int64_t v = f.sg ? (int64_t)(int32_t)f.l[i] : (int64_t)f.l[i];
        int64_t lim = (int64_t)f.lb * (int64_t)(MASK29);
        if (v > lim || v < (f.sg ? -lim : 0)) { std::fprintf(stderr, "BN_BOUNDS: limb %d = %lld exceeds claimed lb %u in %s\n", i, (long long)v, f.lb, where); std::abort(); }
    }
    // value: normalise sum l_i 2^(29 i) into 29-bit digits with a signed carry
    int64_t d[10]; int64_t c = 0;
    for (int i = 0; i < 9; ++i) { int64_t t = (f.sg ? (int64_t)(int32_t)f.l[i] : (int64_t)f.l[i]) + c; if (i < 8) { d[i] = t & MASK29; c = t >> 29; } else { d[8] = t; } }
    bool neg = d[8] < 0;
    if (neg && !f.sg) { std::fprintf(stderr, "BN_BOUNDS: negative value in an unsigned form in %s\n", where); std::abort(); }
    if (neg) {          // magnitude: negate the digit string
        int64_t b = 0;
        for (int i = 0; i < 8; ++i) { int64_t t = -d[i] + b; d[i] = t & MASK29; b = t >> 29; }
        d[8] = -d[8] + b;
    }
    // compare with vb * q
    int64_t q[9]; uint64_t cy = 0;
    for (int i = 0; i < 9; ++i) { uint64_t t = (uint64_t)k::Q[i] * f.vb + cy; if (i < 8) { q[i] = (int64_t)(t & MASK29); cy = t >> 29; } else { q[8] = (int64_t)t; } }
    for (int i = 8; i >= 0; --i) {
        if (d[i] < q[i]) return;
        if (d[i] > q[i]) { std::fprintf(stderr, "BN_BOUNDS: |value| exceeds claimed vb %u (x q) in %s\n", f.vb, where); std::abort(); }
    }
    // |value| == vb*q exactly is allowed (e.g. K*q - 0): every bound argument in this file holds with <= as well
}
#define BN_VERIFY(x, where) bn_verify_actual((x), where)
#else
#define BN_VERIFY(x, where) ((void)0)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Leaf calling convention.  A struct argument larger than the 16 registers clang's AMDGPU ABI grants to ALL aggregate
// arguments together is passed through private memory (scratch), which is HBM traffic (measured: profiles/r01a_*).
// Vectors are not: a <9 x i32> travels in 9 VGPRs.  So every multiplier-sized leaf is a real (noinline) function whose
// operands and result are u32x9 vectors, wrapped by an inline function with the natural Fe signature.
#if defined(BN_HOSTSIM)
#define BN_LEAF1(NAME, BODY) BN_FN Fe NAME(const Fe &a) { return BODY(a); }
#define BN_LEAF1M(NAME, BODY) BN_FN Fe NAME(const Fe &a) { return BODY(a); }
#define BN_LEAF2(NAME, BODY) BN_FN Fe NAME(const Fe &a, const Fe &b) { return BODY(a, b); }
#define BN_LEAF3T(NAME, BODY)                                                                     \
    template <int C1, int C2, int C3> BN_FN Fe NAME(const Fe &a, const Fe &b, const Fe &c) { return BODY<C1, C2, C3>(a, b, c); }
#else
typedef uint32_t u32x9 __attribute__((ext_vector_type(9)));
BN_FN Fe bn_unv(u32x9 v) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = v[i];
    return r;
}
BN_FN u32x9 bn_tov(const Fe &f) {
    u32x9 v;
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = f.l[i];
    return v;
}
#define BN_LEAF1(NAME, BODY)                                                                      \
    BN_LEAF u32x9 NAME##_leaf(u32x9 a) { return bn_tov(BODY(bn_unv(a))); }                       \
    BN_FN Fe NAME(const Fe &a) { return bn_unv(NAME##_leaf(bn_tov(a))); }
#define BN_LEAF1M(NAME, BODY)                                                                     \
    BN_LEAF_MUL u32x9 NAME##_leaf(u32x9 a) { return bn_tov(BODY(bn_unv(a))); }                   \
    BN_FN Fe NAME(const Fe &a) { return bn_unv(NAME##_leaf(bn_tov(a))); }
#define BN_LEAF2(NAME, BODY)                                                                      \
    BN_LEAF_MUL u32x9 NAME##_leaf(u32x9 a, u32x9 b) { return bn_tov(BODY(bn_unv(a), bn_unv(b))); }   \
    BN_FN Fe NAME(const Fe &a, const Fe &b) { return bn_unv(NAME##_leaf(bn_tov(a), bn_tov(b))); }
#define BN_LEAF3T(NAME, BODY)                                                                     \
    template <int C1, int C2, int C3> BN_LEAF_RED u32x9 NAME##_leaf(u32x9 a, u32x9 b, u32x9 c) {     \
        return bn_tov(BODY<C1, C2, C3>(bn_unv(a), bn_unv(b), bn_unv(c)));                         \
    }                                                                                             \
    template <int C1, int C2, int C3> BN_FN Fe NAME(const Fe &a, const Fe &b, const Fe &c) {      \
        return bn_unv(NAME##_leaf<C1, C2, C3>(bn_tov(a), bn_tov(b), bn_tov(c)));                  \
    }
#endif

// ---------------------------------------------------------------------------------------------------------------------
// constants as Fe
template <class T>
BN_FN Fe fe_const(const T &tab) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = tab[i];
    BN_SETB(r, 1, 1);
    return r;
}
BN_FN Fe fe_zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    BN_SETB(r, 1, 1);
    return r;
}
BN_FN Fe fe_one() { return fe_const(k::ONE); }

// K*q written with every low limb raised by LB*(2^29-1)-ish so that  a + Bias - b  never borrows when b has limb bound LB
// and value < (K-1)*q.   Sum of the limbs (weighted) is exactly K*q.
template <int LB, int K>
struct Bias {
    uint32_t c[9];
    constexpr Bias() : c{} {
        uint64_t carry = 0;
        uint32_t n[9] = {};
        for (int i = 0; i < 9; ++i) {
            uint64_t t = (uint64_t)k::Q[i] * (uint64_t)K + carry;
            if (i < 8) { n[i] = (uint32_t)(t & MASK29); carry = t >> 29; } else { n[i] = (uint32_t)t; }
        }
        c[0] = n[0] + ((uint32_t)LB << 29);
        for (int i = 1; i < 8; ++i) c[i] = n[i] + ((uint32_t)LB << 29) - (uint32_t)LB;
        c[8] = n[8] - (uint32_t)LB;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// lazy additive ops (arith.rs:238-253,266-273 do these with a conditional correction per call; here: none)
BN_FN Fe fe_add(const Fe &a, const Fe &b) {
    BN_COUNT(addsub);
    BN_REQUIRE(a.lb + b.lb <= ((a.sg || b.sg) ? 4u : 8u), "fe_add limb overflow");
    BN_REQUIRE(a.vb + b.vb <= 1024, "fe_add value overflow");
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    BN_IFB(bool sg_ = a.sg || b.sg;)
    BN_SETB(r, a.lb + b.lb, a.vb + b.vb);
    BN_IFB(r.sg = sg_;)
    BN_VERIFY(r, "fe_add");
    return r;
}
BN_FN Fe fe_dbl(const Fe &a) { return fe_add(a, a); }
// signed lazy difference: plain limb-wise a - b, limbs become int32 (two's complement), no bias and no carries.  Only
// fe_lc3 may consume the result (it interprets every input limb as a signed 32-bit integer).
BN_FN Fe fe_ssub(const Fe &a, const Fe &b) {
    BN_COUNT(addsub);
    BN_REQUIRE(a.lb + b.lb <= 4, "fe_ssub limb overflow (|limb| must stay < 2^31)");
    BN_REQUIRE(a.vb + b.vb <= 1024, "fe_ssub value overflow");
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] - b.l[i];
    BN_SETB(r, a.lb + b.lb, a.vb + b.vb);
    BN_IFB(r.sg = true;)
    BN_VERIFY(r, "fe_ssub");
    return r;
}

// the difference of two elements with normalized limbs as a signed operand of fe_mul2s: |limb| < 2^29, |value| < max(a, b)
BN_FN Fe fe_sdiff(const Fe &a, const Fe &b) {
    BN_COUNT(addsub);
    BN_REQUIRE(!a.sg && !b.sg && a.lb == 1 && b.lb == 1, "fe_sdiff takes normalized unsigned limbs");
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] - b.l[i];
    BN_SETB(r, 1, (a.vb > b.vb ? a.vb : b.vb));
    BN_IFB(r.sg = true;)
    BN_VERIFY(r, "fe_sdiff");
    return r;
}
// limb-wise negation of a signed (or unsigned) operand, no bias
BN_FN Fe fe_sneg(const Fe &a) {
    BN_COUNT(addsub);
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0u - a.l[i];
    BN_SETB(r, a.lb, a.vb);
    BN_IFB(r.sg = true;)
    BN_VERIFY(r, "fe_sneg");
    return r;
}

// a - b (mod q) as a + K*q - b, b must satisfy lb <= LB and vb <= K-1
template <int LB, int K>
BN_FN Fe fe_sub(const Fe &a, const Fe &b) {
    BN_COUNT(addsub);
    BN_REQUIRE(!a.sg && !b.sg, "fe_sub on a signed lazy value");
    BN_REQUIRE(b.lb <= (uint32_t)LB, "fe_sub: subtrahend limbs exceed the bias");
    BN_REQUIRE(b.vb + 1 <= (uint32_t)K, "fe_sub: subtrahend value exceeds the bias");
    BN_REQUIRE(a.lb + LB + 1 <= 8, "fe_sub limb overflow");
    BN_REQUIRE(a.vb + K <= 1024, "fe_sub value overflow");
    constexpr Bias<LB, K> B{};
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + B.c[i] - b.l[i];
    BN_SETB(r, a.lb + LB + 1, a.vb + K);
    BN_VERIFY(r, "fe_sub");
    return r;
}
template <int LB, int K>
BN_FN Fe fe_neg(const Fe &b) {
    BN_COUNT(addsub);
    BN_REQUIRE(!b.sg, "fe_neg on a signed lazy value");
    BN_REQUIRE(b.lb <= (uint32_t)LB, "fe_neg: limbs exceed the bias");
    BN_REQUIRE(b.vb + 1 <= (uint32_t)K, "fe_neg: value exceeds the bias");
    constexpr Bias<LB, K> B{};
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = B.c[i] - b.l[i];
    BN_SETB(r, LB + 1, K);
    BN_VERIFY(r, "fe_neg");
    return r;
}

// carry propagation only: limbs back to 29 bits, value unchanged
BN_FN Fe fe_norm(const Fe &a) {
    BN_COUNT(norm);
    BN_REQUIRE(!a.sg, "fe_norm on a signed lazy value");
    Fe r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t t = a.l[i] + c;      // a.l[i] <= 8*(2^29-1), c < 2^4: no overflow
        r.l[i] = t & MASK29;
        c = t >> 29;
    }
    r.l[8] = a.l[8] + c;
    BN_SETB(r, 1, a.vb);
    BN_VERIFY(r, "fe_norm");
    return r;
}

// a / 2 (mod q) without a multiplication: if the value is odd add q (q is odd), then shift the whole limb string right by one
// bit.  The parity of the value is the parity of limb 0 whatever the limb bounds (every higher limb weighs a multiple of 2^29).
// Replaces the two products by 2^-1 of the doubling step (groups/mod.rs:615,619: `* two_inv`): ~40 plain instructions instead
// of a 171-multiplication Montgomery product.  Accepts lazy unsigned input; result lb = ceil((lb+2)/2), vb = ceil((vb+1)/2).
BN_FN Fe fe_half(const Fe &a) {
    BN_COUNT(addsub);
    BN_REQUIRE(!a.sg && a.lb <= 6, "fe_half input");
    const uint32_t mask = 0u - (a.l[0] & 1u);
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = a.l[i] + (k::Q[i] & mask);          // <= (lb+1) * (2^29 - 1): fits for lb <= 6
    Fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = (t[i] >> 1) + ((t[i + 1] & 1u) << 28);
    r.l[8] = t[8] >> 1;
    BN_SETB(r, (a.lb + 3) / 2, (a.vb + 2) / 2);
    BN_VERIFY(r, "fe_half");
    return r;
}

// quotient estimate in fe_reduce: k = floor(top * FE_MU / 2^53), FE_MU = floor(2^285/q), top ~ value >> 232
// value reduction: returns the same residue with normalized limbs and value < 2q.   Accepts any lb <= 8, vb <= 1000.
// `scale` multiplies the input by a small constant first (1, 9, ...): reduce(scale * a).
template <int SCALE = 1>
BN_FN Fe fe_reduce(const Fe &a) {
    BN_REQUIRE(a.lb <= 8 && !a.sg, "fe_reduce lb");
    BN_REQUIRE((uint64_t)a.vb * SCALE <= 1000, "fe_reduce vb");
    // top ~ floor(SCALE*value / 2^232), never above it (carries still parked in lower limbs are ignored: at most ~8*SCALE)
    uint32_t top = (uint32_t)SCALE * (a.l[8] + (a.l[7] >> 29));
    uint32_t kq = (uint32_t)(((uint64_t)top * k::FE_MU) >> 53);     // k <= floor(SCALE*value/q), k >= that - 1
    Fe r;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int64_t t = carry + (int64_t)((uint64_t)a.l[i] * (uint32_t)SCALE) - (int64_t)((uint64_t)kq * k::Q[i]);
        if (i < 8) { r.l[i] = (uint32_t)t & MASK29; carry = t >> 29; } else { r.l[i] = (uint32_t)t; }
    }
    BN_SETB(r, 1, 2);
    BN_VERIFY(r, "fe_reduce");
    return r;
}

// reduce(C1*x + C2*y + C3*z) for small signed integer constants (0 disables a term): one signed 64-bit carry chain that
// forms the linear combination, subtracts floor-ish(value/q)*q and renormalizes.  This is how every "sum of products"
// of the tower (Karatsuba recombination, multiplication by xi = 9+i) gets back to standard form; the reference spends a
// conditional add/subtract per Fq add instead (arith.rs:238-253).  Result: normalized limbs, value < 3q.
// core: the middle term's sign is a per-lane run-time flag (needed by the lane-pair Fq2 mapping, where the even lane
// subtracts and the odd lane adds the partner's limb in xi-multiplications)
// WIDE: every term enters the 64-bit chain on its own (no 32-bit pre-combination), which lifts the bound on the SUM of the inputs' limb
// bounds - each input still has to fit int32 (lb <= 4).  Used where one reduction takes the place of several (quad.hpp).
// SIGN2: the middle term's sign really is a per-lane value (the _par callers): the term then enters the 64-bit chain with a per-lane
// coefficient register (one multiply-add per limb) instead of a negate-and-select before the narrow sum (two to three instructions).
template <int C1, int C2, int C3, int C4, bool WIDE = false, bool SIGN2 = false>
BN_FN Fe fe_lc4_core(const Fe &x, const Fe &y, const Fe &z, const Fe &w, bool neg2) {
    BN_COUNT(lc3);
    constexpr int A1 = C1 < 0 ? -C1 : C1, A2 = C2 < 0 ? -C2 : C2, A3 = C3 < 0 ? -C3 : C3, A4 = C4 < 0 ? -C4 : C4;
    // Every input limb is read as a SIGNED 32-bit integer (so inputs may be signed lazy differences, fe_ssub): |limb| < 2^31,
    // i.e. lb <= 4.  Terms with a small coefficient are first combined in 32-bit arithmetic ("narrow"); the others enter the
    // 64-bit chain on their own.
    // A LONE small term with a coefficient other than +1 would cost a shift or a negation AND the multiply-add that takes the narrow
    // sum into the chain: it goes into the chain directly instead (one multiply-add).
    constexpr bool K1 = !WIDE && C1 != 0 && A1 <= 2, K2 = !WIDE && !SIGN2 && C2 != 0 && A2 <= 2, K3 = !WIDE && C3 != 0 && A3 <= 2, K4 = !WIDE && C4 != 0 && A4 <= 2;
    constexpr bool LONE = (int)K1 + (int)K2 + (int)K3 + (int)K4 == 1;
    constexpr bool N1 = K1 && !(LONE && C1 != 1), N2 = K2 && !(LONE && C2 != 1), N3 = K3 && !(LONE && C3 != 1), N4 = K4 && !(LONE && C4 != 1);
    BN_COUNT_MACS(1 + 9 * (1 + (C1 != 0 && !N1) + (C2 != 0 && !N2) + (C3 != 0 && !N3) + (C4 != 0 && !N4) + ((N1 || N2 || N3 || N4) ? 1 : 0)));   // v_mul_hi_i32 (quotient), then per limb: kq x (-q_i), one per wide term, one for the narrow sum
    BN_IFB(if (!((C1 == 0 || x.lb <= 4) && (C2 == 0 || y.lb <= 4) && (C3 == 0 || z.lb <= 4) && (C4 == 0 || w.lb <= 4)))
               std::fprintf(stderr, "fe_lc<%d,%d,%d,%d> lbs %u %u %u %u\n", C1, C2, C3, C4, x.lb, y.lb, z.lb, w.lb);)
    BN_REQUIRE((C1 == 0 || x.lb <= 4) && (C2 == 0 || y.lb <= 4) && (C3 == 0 || z.lb <= 4) && (C4 == 0 || w.lb <= 4), "fe_lc: input limbs must fit int32");
    BN_REQUIRE((uint64_t)A1 * x.vb + (uint64_t)A2 * y.vb + (uint64_t)A3 * z.vb + (uint64_t)A4 * w.vb <= 500, "fe_lc vb");     // keeps the quotient estimate below in 32 bits: |te| < 500 * 2^21.6 + 1000 < 2^31
    BN_IFB(if ((N1 ? A1 * x.lb : 0) + (N2 ? A2 * y.lb : 0) + (N3 ? A3 * z.lb : 0) + (N4 ? A4 * w.lb : 0) > 4)
               std::fprintf(stderr, "fe_lc<%d,%d,%d,%d> lbs %u %u %u %u\n", C1, C2, C3, C4, x.lb, y.lb, z.lb, w.lb);)
    BN_REQUIRE((N1 ? A1 * x.lb : 0) + (N2 ? A2 * y.lb : 0) + (N3 ? A3 * z.lb : 0) + (N4 ? A4 * w.lb : 0) <= 4, "fe_lc narrow part exceeds 32 bits");
    // signed estimate of floor(value / 2^232) that never exceeds the truth (margins: carries still parked in lower limbs,
    // at most 9 per unit coefficient, and the truncation of FE_MU24, < 2^9 units).
    // The top limb alone: the margin of 9 per unit coefficient covers what limb 7 holds beyond its 29 bits (|.| <= 4 units for limb bound
    // 4; one unit is 3.2e-7 q).  NOTE for whoever touches the build flags: with this form LLVM's DPP combiner folds the quad_perm move of a
    // neighbour pair's top limb into the subtraction that forms `te` - `v_subrev_u32_dpp d, x, a` where the DPP value is the subtrahend - and on
    // gfx950 that instruction computes dpp(a) - x, not a - dpp(x): the permutation lands on the operand that becomes the minuend after the
    // opcode's operand reversal (round 5, profiles/r05_dpp_fold_bisect.txt: one hand-folded instruction in otherwise good assembly makes every
    // four-lane pairing wrong; tools/variants/dpp_subrev_check.hip shows what it computes).  The library is therefore built with
    // -mllvm -amdgpu-dpp-combine=false in EVERY unit (bn_amd/_native.py; free: profiles/r05_ab_dpp_combine_off.txt), tests/test_build_quality.py
    // rejects any DPP instruction other than v_mov_b32_dpp in the shipped code objects, and a GPU test rebuilds kernel units on the box and
    // re-runs the goldens.
    auto top = [](const Fe &f) -> int32_t { return (int32_t)f.l[8]; };
    const int32_t c2 = neg2 ? -C2 : C2;
    // 32-bit arithmetic (a value below vb q has a top limb below vb * 2^21.6, and the sum of |C| vb is at most 500: checked above):
    // one v_mul_hi_i32 and a shift on the GPU instead of a 64 x 32-bit product
    int32_t te = -600 - 9 * (A1 + A2 + A3 + A4);
    if (C1 != 0) te += C1 * top(x);
    if (C2 != 0) te += c2 * top(y);
    if (C3 != 0) te += C3 * top(z);
    if (C4 != 0) te += C4 * top(w);
    const int64_t kq = ((int64_t)te * (int64_t)k::FE_MU24) >> 45;     // floor; kq <= floor(value/q), kq >= value/q - 2
    Fe r;
    int64_t carry = 0;
#if !defined(BN_HOSTSIM)
    // GPU: the 64-bit chain of a limb as ONE asm statement of v_mad_i64_i32 (carry - kq q_i + narrow sum + the wide terms), like the
    // multiplier leaves (fe_asm.hpp): the compiler otherwise builds it from sign extensions, 64-bit adds and - in some contexts -
    // v_mul_lo / v_mul_hi pairs.  Same arithmetic, limb for limb.
    // kq * (-q_i): the quotient estimate itself against the NEGATED modulus limb (a signed 32-bit scalar operand), so no negation of kq;
    // limb 0 starts its chain from the inline constant 0 (no accumulator to clear).
    const int32_t kq32 = (int32_t)kq;
    constexpr int NWIDE = (C1 != 0 && !N1) + (C2 != 0 && !N2) + (C3 != 0 && !N3) + (C4 != 0 && !N4);
    constexpr bool ANY_NARROW = N1 || N2 || N3 || N4;
#define BN_LC_NARROW "\n\tv_mad_i64_i32 %0, vcc, 1, %3, %0"
#define BN_LC_CHAIN(OUT, ACC0)                                                                                                                        \
    if constexpr (NWIDE == 0) {                                                                                                                       \
        if constexpr (ANY_NARROW) asm("v_mad_i64_i32 %0, vcc, %1, %2, " ACC0 BN_LC_NARROW : OUT(t) : "v"(kq32), "s"(nqi), "v"(nsum) : "vcc");        \
        else asm("v_mad_i64_i32 %0, vcc, %1, %2, " ACC0 : OUT(t) : "v"(kq32), "s"(nqi) : "vcc");                                                      \
    } else if constexpr (NWIDE == 1) {                                                                                                                \
        if constexpr (ANY_NARROW) asm("v_mad_i64_i32 %0, vcc, %1, %2, " ACC0 BN_LC_NARROW "\n\tv_mad_i64_i32 %0, vcc, %4, %5, %0" : OUT(t) : "v"(kq32), "s"(nqi), "v"(nsum), "v"(wc[0]), "v"(wv[0]) : "vcc"); \
        else asm("v_mad_i64_i32 %0, vcc, %1, %2, " ACC0 "\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0" : OUT(t) : "v"(kq32), "s"(nqi), "v"(wc[0]), "v"(wv[0]) : "vcc"); \
    } else if constexpr (NWIDE == 2) {                                                                                                                \
        if constexpr (ANY_NARROW) asm("v_mad_i64_i32 %0, vcc, %1, %2, " ACC0 BN_LC_NARROW "\n\tv_mad_i64_i32 %0, vcc, %4, %5, %0\n\tv_mad_i64_i32 %0, vcc, %6, %7, %0" : OUT(t) : "v"(kq32), "s"(nqi), "v"(nsum), "v"(wc[0]), "v"(wv[0]), "v"(wc[1]), "v"(wv[1]) : "vcc"); \
        else asm("v_mad_i64_i32 %0, vcc, %1, %2, " ACC0 "\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0" : OUT(t) : "v"(kq32), "s"(nqi), "v"(wc[0]), "v"(wv[0]), "v"(wc[1]), "v"(wv[1]) : "vcc"); \
    } else if constexpr (NWIDE == 3) {                                                                                                                \
        if constexpr (ANY_NARROW) asm("v_mad_i64_i32 %0, vcc, %1, %2, " ACC0 BN_LC_NARROW "\n\tv_mad_i64_i32 %0, vcc, %4, %5, %0\n\tv_mad_i64_i32 %0, vcc, %6, %7, %0\n\tv_mad_i64_i32 %0, vcc, %8, %9, %0" : OUT(t) : "v"(kq32), "s"(nqi), "v"(nsum), "v"(wc[0]), "v"(wv[0]), "v"(wc[1]), "v"(wv[1]), "v"(wc[2]), "v"(wv[2]) : "vcc"); \
        else asm("v_mad_i64_i32 %0, vcc, %1, %2, " ACC0 "\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0" : OUT(t) : "v"(kq32), "s"(nqi), "v"(wc[0]), "v"(wv[0]), "v"(wc[1]), "v"(wv[1]), "v"(wc[2]), "v"(wv[2]) : "vcc"); \
    } else {                                                                                                                                          \
        asm("v_mad_i64_i32 %0, vcc, %1, %2, " ACC0 "\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\tv_mad_i64_i32 %0, vcc, %9, %10, %0" : OUT(t) : "v"(kq32), "s"(nqi), "v"(wc[0]), "v"(wv[0]), "v"(wc[1]), "v"(wv[1]), "v"(wc[2]), "v"(wv[2]), "v"(wc[3]), "v"(wv[3]) : "vcc"); \
        if constexpr (ANY_NARROW) t += (int64_t)nsum;                                                                                                 \
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int32_t nsum = 0;
        if (N1) nsum += C1 * (int32_t)x.l[i];
        if (N2) { int32_t yy = C2 * (int32_t)y.l[i]; nsum += neg2 ? -yy : yy; }
        if (N3) nsum += C3 * (int32_t)z.l[i];
        if (N4) nsum += C4 * (int32_t)w.l[i];
        // the wide terms in order
        int32_t wc[4], wv[4];
        int nw = 0;
        if (C1 != 0 && !N1) { wc[nw] = C1; wv[nw] = (int32_t)x.l[i]; ++nw; }
        if (C2 != 0 && !N2) { wc[nw] = c2; wv[nw] = (int32_t)y.l[i]; ++nw; }
        if (C3 != 0 && !N3) { wc[nw] = C3; wv[nw] = (int32_t)z.l[i]; ++nw; }
        if (C4 != 0 && !N4) { wc[nw] = C4; wv[nw] = (int32_t)w.l[i]; ++nw; }
        int64_t t = carry;
        const int32_t nqi = -(int32_t)k::Q[i];
        if (i == 0) { BN_LC_CHAIN("=&v", "0") } else { BN_LC_CHAIN("+v", "%0") }
        if (i < 8) { r.l[i] = (uint32_t)t & MASK29; carry = t >> 29; } else { r.l[i] = (uint32_t)t; }
    }
#undef BN_LC_CHAIN
#undef BN_LC_NARROW
#else
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int32_t nsum = 0;
        if (N1) nsum += C1 * (int32_t)x.l[i];
        if (N2) { int32_t yy = C2 * (int32_t)y.l[i]; nsum += neg2 ? -yy : yy; }
        if (N3) nsum += C3 * (int32_t)z.l[i];
        if (N4) nsum += C4 * (int32_t)w.l[i];
        int64_t t = carry + (int64_t)nsum - kq * (int64_t)k::Q[i];
        if (C1 != 0 && !N1) t += (int64_t)C1 * (int64_t)(int32_t)x.l[i];
        if (C2 != 0 && !N2) t += (int64_t)c2 * (int64_t)(int32_t)y.l[i];
        if (C3 != 0 && !N3) t += (int64_t)C3 * (int64_t)(int32_t)z.l[i];
        if (C4 != 0 && !N4) t += (int64_t)C4 * (int64_t)(int32_t)w.l[i];
        if (i < 8) { r.l[i] = (uint32_t)t & MASK29; carry = t >> 29; } else { r.l[i] = (uint32_t)t; }
    }
#endif
    // floor() loses < 1 and the margins of `te` ~3e-4, so kq > value/q - 1.001 and the result is < 1.001 q
    BN_SETB(r, 1, 2);
    BN_VERIFY(r, "fe_lc4_core");
    return r;
}
template <int C1, int C2, int C3, bool SIGN2 = false>
BN_FN Fe fe_lc3_core(const Fe &x, const Fe &y, const Fe &z, bool neg2) { return fe_lc4_core<C1, C2, C3, 0, false, SIGN2>(x, y, z, z, neg2); }
// all-64-bit variant for UNSIGNED lazy inputs with limbs beyond 31 bits (lb up to 8); rare call sites only
template <int C1, int C2, int C3>
BN_FN Fe fe_lc3w_body(const Fe &x, const Fe &y, const Fe &z) {
    BN_COUNT(lc3w);
    constexpr int A1 = C1 < 0 ? -C1 : C1, A2 = C2 < 0 ? -C2 : C2, A3 = C3 < 0 ? -C3 : C3;
    BN_REQUIRE((C1 == 0 || (x.lb <= 8 && !x.sg)) && (C2 == 0 || (y.lb <= 8 && !y.sg)) && (C3 == 0 || (z.lb <= 8 && !z.sg)), "fe_lc3w lb");
    BN_REQUIRE((uint64_t)A1 * x.vb + (uint64_t)A2 * y.vb + (uint64_t)A3 * z.vb <= 1000, "fe_lc3w vb");
    auto top = [](const Fe &f) -> int64_t { return (int64_t)f.l[8] + (int64_t)(f.l[7] >> 29); };
    int64_t te = -600 - 9 * (int64_t)(A1 + A2 + A3);
    if (C1 != 0) te += (int64_t)C1 * top(x);
    if (C2 != 0) te += (int64_t)C2 * top(y);
    if (C3 != 0) te += (int64_t)C3 * top(z);
    const int64_t kq = (te * (int64_t)k::FE_MU24) >> 45;
    Fe r;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int64_t t = carry - kq * (int64_t)k::Q[i];
        if (C1 != 0) t += (int64_t)C1 * (int64_t)x.l[i];
        if (C2 != 0) t += (int64_t)C2 * (int64_t)y.l[i];
        if (C3 != 0) t += (int64_t)C3 * (int64_t)z.l[i];
        if (i < 8) { r.l[i] = (uint32_t)t & MASK29; carry = t >> 29; } else { r.l[i] = (uint32_t)t; }
    }
    BN_SETB(r, 1, 2);
    BN_VERIFY(r, "fe_lc3w_body");
    return r;
}
BN_LEAF3T(fe_lc3w, fe_lc3w_body)
template <int C1, int C2, int C3>
BN_FN Fe fe_lc3_body(const Fe &x, const Fe &y, const Fe &z) { return fe_lc3_core<C1, C2, C3>(x, y, z, false); }
BN_LEAF3T(fe_lc3, fe_lc3_body)
BN_FN Fe fe_std(const Fe &x) { return fe_lc3<1, 0, 0>(x, x, x); }       // any lazy value -> standard form (1,3)

// ---------------------------------------------------------------------------------------------------------------------
// Montgomery multiplication, product scanning with interleaved reduction (replaces arith.rs:481-503 mul_reduce +
// :257-263).  One 64-bit accumulator per column, 81 + 81 v_mad_u64_u32, no carry instructions.
// Column bound: 9*(la*lb) * 2^58 + 9 * 2^58 + carry < 2^64  <=>  la*lb <= 6.
// Value: result < (A*B/169.3 + 1) q, so A*B <= 169 gives < 2q.
// On the GPU the three multiplier leaves below run the same arithmetic with every column's multiply-add chain as ONE inline-asm
// statement (fe_asm.hpp, tools/gen_asm_leaf.py): the compiler otherwise splits each column sum into two chains and joins them with
// a 64-bit add (16 v_lshl_add_u64 per product); +3.6 % pairings/s (profiles/r03_ab_asm_leaf.txt).  The host simulation uses the C++
// bodies, which also carry the bound checks.
#if !defined(BN_HOSTSIM)
}  // namespace bn254
#include "fe_asm.hpp"
namespace bn254 {
#endif
BN_FN Fe fe_mul_body(const Fe &a, const Fe &b) {
#if !defined(BN_HOSTSIM)
    return fe_mul_asm(a, b);
#endif
    BN_COUNT(mul); BN_COUNT_MACS(171);
    BN_REQUIRE(!a.sg && !b.sg, "fe_mul on a signed lazy value");
    BN_REQUIRE(a.lb * b.lb <= 6, "fe_mul column overflow");
    BN_REQUIRE(a.vb * b.vb <= 169, "fe_mul value bound");
    uint64_t acc = 0;
    uint32_t m[9];
    Fe r;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
#pragma unroll
        for (int i = 0; i <= c; ++i) acc += (uint64_t)a.l[i] * b.l[c - i];
#pragma unroll
        for (int i = 0; i < c; ++i) acc += (uint64_t)m[i] * k::Q[c - i];
        m[c] = ((uint32_t)acc * k::QINV) & MASK29;
        acc += (uint64_t)m[c] * k::Q[0];
        acc >>= 29;
    }
#pragma unroll
    for (int c = 9; c < 17; ++c) {
#pragma unroll
        for (int i = c - 8; i <= 8; ++i) acc += (uint64_t)a.l[i] * b.l[c - i];
#pragma unroll
        for (int i = c - 8; i <= 8; ++i) acc += (uint64_t)m[i] * k::Q[c - i];
        r.l[c - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    BN_SETB(r, 1, 2);
    BN_VERIFY(r, "fe_mul_body");
    return r;
}
BN_LEAF2(fe_mul, fe_mul_body)
// a*a / R: the 36 cross products a_i a_j (i < j) are taken once against the doubled limbs 2 a_j, so the product part is 45
// instead of 81 mads (+ the same 81 of the reduction).  Same column sums as fe_mul(a, a), hence the same bounds.
BN_FN Fe fe_sqr_body(const Fe &a) {
#if !defined(BN_HOSTSIM)
    return fe_sqr_asm(a);
#endif
    BN_COUNT(mul); BN_COUNT_MACS(45 + 81 + 9);
    BN_REQUIRE(!a.sg, "fe_sqr on a signed lazy value");
    BN_REQUIRE(a.lb * a.lb <= 6, "fe_sqr column overflow");
    BN_REQUIRE(a.vb * a.vb <= 169, "fe_sqr value bound");
    uint32_t a2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a2[i] = a.l[i] << 1;
    uint64_t acc = 0;
    uint32_t m[9];
    Fe r;
#pragma unroll
    for (int c = 0; c < 17; ++c) {
        const int lo = c < 9 ? 0 : c - 8, hi = c < 9 ? c : 8;
#pragma unroll
        for (int i = lo; i <= hi; ++i) {
            const int j = c - i;
            if (i < j) acc += (uint64_t)a.l[i] * a2[j];
            else if (i == j) acc += (uint64_t)a.l[i] * a.l[i];
        }
        if (c < 9) {
#pragma unroll
            for (int i = 0; i < c; ++i) acc += (uint64_t)m[i] * k::Q[c - i];
            m[c] = ((uint32_t)acc * k::QINV) & MASK29;
            acc += (uint64_t)m[c] * k::Q[0];
        } else {
#pragma unroll
            for (int i = c - 8; i <= 8; ++i) acc += (uint64_t)m[i] * k::Q[c - i];
            r.l[c - 9] = (uint32_t)acc & MASK29;
        }
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    BN_SETB(r, 1, 2);
    BN_VERIFY(r, "fe_sqr_body");
    return r;
}
BN_LEAF1M(fe_sqr, fe_sqr_body)

// (a*u + c*v) / R with ONE reduction: 162 + 81 mads.   Column bound: la*lu + lc*lv <= 6; value: A*U + C*V <= 169.
BN_FN Fe fe_mul2(const Fe &a, const Fe &u, const Fe &c, const Fe &v) {
#if !defined(BN_HOSTSIM)
    return fe_mul2_asm(a, u, c, v);
#endif
    BN_COUNT(mul2); BN_COUNT_MACS(252);
    BN_REQUIRE(!a.sg && !u.sg && !c.sg && !v.sg, "fe_mul2 on a signed lazy value");
    BN_REQUIRE(a.lb * u.lb + c.lb * v.lb <= 6, "fe_mul2 column overflow");
    BN_REQUIRE(a.vb * u.vb + c.vb * v.vb <= 338, "fe_mul2 value bound");           // <= 169: result < 2q;  <= 338: result < 3q
    uint64_t acc = 0;
    uint32_t m[9];
    Fe r;
#pragma unroll
    for (int col = 0; col < 9; ++col) {
#pragma unroll
        for (int i = 0; i <= col; ++i) {
            acc += (uint64_t)a.l[i] * u.l[col - i];
            acc += (uint64_t)c.l[i] * v.l[col - i];
        }
#pragma unroll
        for (int i = 0; i < col; ++i) acc += (uint64_t)m[i] * k::Q[col - i];
        m[col] = ((uint32_t)acc * k::QINV) & MASK29;
        acc += (uint64_t)m[col] * k::Q[0];
        acc >>= 29;
    }
#pragma unroll
    for (int col = 9; col < 17; ++col) {
#pragma unroll
        for (int i = col - 8; i <= 8; ++i) {
            acc += (uint64_t)a.l[i] * u.l[col - i];
            acc += (uint64_t)c.l[i] * v.l[col - i];
        }
#pragma unroll
        for (int i = col - 8; i <= 8; ++i) acc += (uint64_t)m[i] * k::Q[col - i];
        r.l[col - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    BN_SETB(r, 1, (a.vb * u.vb + c.vb * v.vb <= 169 ? 2 : 3));
    BN_VERIFY(r, "fe_mul2");
    return r;
}

// The dual product over SIGNED operands: every limb an int32 of magnitude below 2^29 - differences of two standard elements (fe_sdiff),
// where a Karatsuba cross product would otherwise take sums and pay a carry propagation to get one of them back under 2^29 (tower.hpp
// f6_mul).  27 terms of magnitude below 2^58 stay inside a signed 64-bit column; the carries are arithmetic shifts; the Montgomery digit
// comes from the low 29 bits of the two's complement column as before.  The result is a SIGNED lazy value: limbs 0..7 in [0, 2^29), the
// top limb carries the sign, |value| < ((A U + C V) / 169.3 + 1) q.  Only the fused reductions (fe_lc*) consume it.
BN_FN Fe fe_mul2s(const Fe &a, const Fe &u, const Fe &c, const Fe &v) {
#if !defined(BN_HOSTSIM)
    return fe_mul2s_asm(a, u, c, v);
#endif
    BN_COUNT(mul2); BN_COUNT_MACS(252);
    BN_REQUIRE(a.lb * u.lb + c.lb * v.lb <= 2, "fe_mul2s column overflow (signed columns hold 27 terms of 2^58)");
    BN_REQUIRE(a.vb * u.vb + c.vb * v.vb <= 169, "fe_mul2s value bound");
    int64_t acc = 0;
    uint32_t m[9];
    Fe r;
#pragma unroll
    for (int col = 0; col < 9; ++col) {
#pragma unroll
        for (int i = 0; i <= col; ++i) {
            acc += (int64_t)(int32_t)a.l[i] * (int64_t)(int32_t)u.l[col - i];
            acc += (int64_t)(int32_t)c.l[i] * (int64_t)(int32_t)v.l[col - i];
        }
#pragma unroll
        for (int i = 0; i < col; ++i) acc += (int64_t)((uint64_t)m[i] * k::Q[col - i]);
        m[col] = ((uint32_t)acc * k::QINV) & MASK29;
        acc += (int64_t)((uint64_t)m[col] * k::Q[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int col = 9; col < 17; ++col) {
#pragma unroll
        for (int i = col - 8; i <= 8; ++i) {
            acc += (int64_t)(int32_t)a.l[i] * (int64_t)(int32_t)u.l[col - i];
            acc += (int64_t)(int32_t)c.l[i] * (int64_t)(int32_t)v.l[col - i];
        }
#pragma unroll
        for (int i = col - 8; i <= 8; ++i) acc += (int64_t)((uint64_t)m[i] * k::Q[col - i]);
        r.l[col - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    BN_SETB(r, 1, 2);
    BN_IFB(r.sg = true;)
    BN_VERIFY(r, "fe_mul2s");
    return r;
}
// (a1 u1 + c1 v1 + a2 u2 + c2 v2 + a3 u3 + c3 v3) / R with ONE reduction: 486 + 81 mads.  Three Fq2 products of the lane-pair mapping summed
// before they are reduced (tower.hpp f12_mul_by_024: every output coefficient of the sparse line product is such a sum).
// Column bound: 6 * 9 + 9 = 63 terms of (2^29 - 1)^2 plus a carry stay below 2^64 ONLY for normalized limbs - every operand must have lb = 1.
// Value: sum of the vb products <= 338 gives a result below 3q.
BN_FN Fe fe_mul6(const Fe &a1, const Fe &u1, const Fe &c1, const Fe &v1, const Fe &a2, const Fe &u2, const Fe &c2, const Fe &v2,
                 const Fe &a3, const Fe &u3, const Fe &c3, const Fe &v3) {
#if !defined(BN_HOSTSIM)
    return fe_mul6_asm(a1, u1, c1, v1, a2, u2, c2, v2, a3, u3, c3, v3);
#endif
    BN_COUNT(mul2); BN_COUNT(mul2); BN_COUNT(mul2);           // (counted as three dual products: the executed-chain figures stay comparable)
    BN_COUNT_MACS(486 + 81 + 9);
    const Fe *x[6] = {&a1, &c1, &a2, &c2, &a3, &c3}, *y[6] = {&u1, &v1, &u2, &v2, &u3, &v3};
#if defined(BN_BOUNDS)
    unsigned vsum = 0;
    for (int k = 0; k < 6; ++k) {
        BN_REQUIRE(!x[k]->sg && !y[k]->sg, "fe_mul6 on a signed lazy value");
        BN_REQUIRE(x[k]->lb == 1 && y[k]->lb == 1, "fe_mul6 needs normalized limbs in every operand (63 column terms)");
        vsum += x[k]->vb * y[k]->vb;
    }
    BN_REQUIRE(vsum <= 338, "fe_mul6 value bound");                       // <= 169: result < 2q;  <= 338: result < 3q
#endif
    uint64_t acc = 0;
    uint32_t m[9];
    Fe r;
#pragma unroll
    for (int col = 0; col < 9; ++col) {
#pragma unroll
        for (int i = 0; i <= col; ++i)
#pragma unroll
            for (int k = 0; k < 6; ++k) acc += (uint64_t)x[k]->l[i] * y[k]->l[col - i];
#pragma unroll
        for (int i = 0; i < col; ++i) acc += (uint64_t)m[i] * k::Q[col - i];
        m[col] = ((uint32_t)acc * k::QINV) & MASK29;
        acc += (uint64_t)m[col] * k::Q[0];
        acc >>= 29;
    }
#pragma unroll
    for (int col = 9; col < 17; ++col) {
#pragma unroll
        for (int i = col - 8; i <= 8; ++i)
#pragma unroll
            for (int k = 0; k < 6; ++k) acc += (uint64_t)x[k]->l[i] * y[k]->l[col - i];
#pragma unroll
        for (int i = col - 8; i <= 8; ++i) acc += (uint64_t)m[i] * k::Q[col - i];
        r.l[col - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    BN_SETB(r, 1, (vsum <= 169 ? 2 : 3));
    BN_VERIFY(r, "fe_mul6");
    return r;
}

// (a1 u1 + c1 v1 + a2 u2 + c2 v2 + a3 u3) / R with ONE reduction: 405 + 81 mads.  Two Fq2 products of the lane-pair mapping plus ONE product of
// this lane's component by a scalar from Fq (replicated in both lanes of the pair): an output coefficient of the NATIVE prepared line
// product (tower.hpp f12_mul_by_line_native), whose v w coefficient is y_P / x_P in Fq.  Same column bound as fe_mul6 (54 terms here).
BN_FN Fe fe_mul5(const Fe &a1, const Fe &u1, const Fe &c1, const Fe &v1, const Fe &a2, const Fe &u2, const Fe &c2, const Fe &v2,
                 const Fe &a3, const Fe &u3) {
#if !defined(BN_HOSTSIM)
    return fe_mul5_asm(a1, u1, c1, v1, a2, u2, c2, v2, a3, u3);
#endif
    BN_COUNT(mul2); BN_COUNT(mul2); BN_COUNT(mul);
    BN_COUNT_MACS(405 + 81 + 9);
    const Fe *x[5] = {&a1, &c1, &a2, &c2, &a3}, *y[5] = {&u1, &v1, &u2, &v2, &u3};
#if defined(BN_BOUNDS)
    unsigned vsum = 0;
    for (int k = 0; k < 5; ++k) {
        BN_REQUIRE(!x[k]->sg && !y[k]->sg, "fe_mul5 on a signed lazy value");
        BN_REQUIRE(x[k]->lb == 1 && y[k]->lb == 1, "fe_mul5 needs normalized limbs in every operand (54 column terms)");
        vsum += x[k]->vb * y[k]->vb;
    }
    BN_REQUIRE(vsum <= 338, "fe_mul5 value bound");                       // <= 169: result < 2q;  <= 338: result < 3q
#endif
    uint64_t acc = 0;
    uint32_t m[9];
    Fe r;
#pragma unroll
    for (int col = 0; col < 9; ++col) {
#pragma unroll
        for (int i = 0; i <= col; ++i)
#pragma unroll
            for (int k = 0; k < 5; ++k) acc += (uint64_t)x[k]->l[i] * y[k]->l[col - i];
#pragma unroll
        for (int i = 0; i < col; ++i) acc += (uint64_t)m[i] * k::Q[col - i];
        m[col] = ((uint32_t)acc * k::QINV) & MASK29;
        acc += (uint64_t)m[col] * k::Q[0];
        acc >>= 29;
    }
#pragma unroll
    for (int col = 9; col < 17; ++col) {
#pragma unroll
        for (int i = col - 8; i <= 8; ++i)
#pragma unroll
            for (int k = 0; k < 5; ++k) acc += (uint64_t)x[k]->l[i] * y[k]->l[col - i];
#pragma unroll
        for (int i = col - 8; i <= 8; ++i) acc += (uint64_t)m[i] * k::Q[col - i];
        r.l[col - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    BN_SETB(r, 1, (vsum <= 169 ? 2 : 3));
    BN_VERIFY(r, "fe_mul5");
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// boundary conversions: the C ABI speaks the reference's format (8 x u32 = [u64;4] little endian, a*2^256 mod q, < q)
BN_FN Fe fe_unpack_u32x8(const uint32_t *w) {     // raw 256-bit integer -> 29-bit limbs (no Montgomery change)
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)w[wi] | (wi + 1 < 8 ? ((uint64_t)w[wi + 1] << 32) : 0);
        r.l[i] = (uint32_t)(two >> sh) & (i < 8 ? MASK29 : 0xffffffffu);
    }
    BN_SETB(r, 1, 1);
    return r;
}
BN_FN void fe_pack_u32x8(const Fe &a, uint32_t *w) {   // normalized limbs, value < 2^256 -> 8 words
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int bit = 32 * j, li = bit / 29, sh = bit - 29 * li;      // word j starts inside limb li at offset sh
        uint64_t v = (uint64_t)a.l[li] >> sh;
        int have = 29 - sh;
        if (li + 1 < 9) v |= (uint64_t)a.l[li + 1] << have;
        if (have + 29 < 32 && li + 2 < 9) v |= (uint64_t)a.l[li + 2] << (have + 29);
        w[j] = (uint32_t)v;
    }
}
// reference image (canonical, radix 2^256) -> internal (radix 2^261), result (1,2)
BN_FN Fe fe_from_u32x8(const uint32_t *w) { return fe_mul(fe_unpack_u32x8(w), fe_const(k::C_IN)); }

// exact canonical value of a (1,2) element: conditional subtraction of q
BN_FN Fe fe_canonical(const Fe &t) {
    BN_REQUIRE(t.lb == 1 && t.vb <= 2, "fe_canonical expects a Montgomery product");
    Fe d;
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int32_t s = (int32_t)t.l[i] - (int32_t)k::Q[i] + borrow;      // |s| < 2^30
        if (i < 8) { d.l[i] = (uint32_t)s & MASK29; borrow = s >> 29; } else { d.l[i] = (uint32_t)s; borrow = s >> 31; }
    }
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = borrow ? t.l[i] : d.l[i];
    BN_SETB(r, 1, 1);
    BN_VERIFY(r, "fe_canonical");
    return r;
}
// internal -> reference image, exact (this is where "bit-exact vs the reference" is decided)
BN_FN void fe_to_u32x8(const Fe &a, uint32_t *w) {
    Fe t = fe_canonical(fe_mul(a, fe_const(k::C_OUT)));
    fe_pack_u32x8(t, w);
}
// a == 0 (mod q) for any lazy a with lb*1 <= 6, vb <= 169
BN_FN bool fe_is_zero(const Fe &a) {
    Fe t = fe_canonical(fe_mul(a, fe_const(k::RAW_ONE)));
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o |= t.l[i];
    return o == 0;
}
// a == 0 (mod q) for a NORMALIZED value below 4q (every product, fused reduction and carry-propagated sum of two of them): the
// value is 0 exactly when it equals 0, q, 2q or 3q, so four limb-wise comparisons replace the Montgomery product +
// canonicalisation of fe_is_zero.  (The equal-point test of every group addition runs this twice.)
struct QMultiples {
    uint32_t m[4][9];
    constexpr QMultiples() : m{} {
        for (int kk = 0; kk < 4; ++kk) {
            uint64_t carry = 0;
            for (int i = 0; i < 9; ++i) {
                uint64_t t = (uint64_t)k::Q[i] * (uint64_t)kk + carry;
                if (i < 8) { m[kk][i] = (uint32_t)(t & MASK29); carry = t >> 29; } else { m[kk][i] = (uint32_t)t; }
            }
        }
    }
};
BN_FN bool fe_is_zero_std(const Fe &a) {
    BN_REQUIRE(!a.sg && a.lb == 1 && a.vb <= 4, "fe_is_zero_std expects normalized limbs and a value below 4q");
    constexpr QMultiples QM{};
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { d0 |= a.l[i]; d1 |= a.l[i] ^ QM.m[1][i]; d2 |= a.l[i] ^ QM.m[2][i]; d3 |= a.l[i] ^ QM.m[3][i]; }
    return (d0 == 0) | (d1 == 0) | (d2 == 0) | (d3 == 0);
}
// per-lane select without divergence
BN_FN Fe fe_select(bool take_b, const Fe &a, const Fe &b) {
    BN_COUNT(select);
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = take_b ? b.l[i] : a.l[i];
    BN_IFB(r.lb = a.lb > b.lb ? a.lb : b.lb; r.vb = a.vb > b.vb ? a.vb : b.vb;)
    return r;
}

// a^(q-2) (Fermat): 362 Montgomery products.  Kept as the independent cross-check of the divsteps inversion below
// (tests/hostsim) - both are uniform across lanes, unlike the reference's data-dependent binary EEA (arith.rs:281-327).
BN_FN Fe fe_inverse_fermat(const Fe &a_in) {
    Fe a = a_in;
    BN_REQUIRE(a.lb <= 2 && a.vb <= 8, "fe_inverse input");
    Fe r = fe_one();
#pragma unroll 1
    for (int i = 253; i >= 0; --i) {
        r = fe_sqr(r);
        if ((k::Q_MINUS_2[i >> 6] >> (i & 63)) & 1) r = fe_mul(r, a);     // exponent bits are wave-uniform: scalar branch
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// Inversion by Bernstein-Yang "safegcd" division steps (the constant-time modular inversion of libsecp256k1's modinv32,
// restated for this modulus): 20 rounds x 30 divsteps = 600 >= the 590 that provably suffice for a 256-bit modulus.  Every
// lane executes exactly the same instruction sequence whatever its data: ~15 k plain 32/64-bit integer operations instead
// of the 362 Montgomery products (~95 k instructions) of the Fermat chain.  The modular inverse is unique, so the bytes that
// leave the engine are the reference's (fp.rs:103-112); inverse(0) = 0.
struct Signed30 { int32_t v[9]; };             // value = sum v[i] * 2^(30 i), limbs in (-2^30, 2^30)

BN_FN int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t &tu, int32_t &tv, int32_t &tq, int32_t &tr) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll 1
    for (int i = 0; i < 30; ++i) {
        uint32_t mask1 = (uint32_t)(zeta >> 31);           // zeta < 0
        uint32_t mask2 = 0u - (g & 1u);                    // g odd
        uint32_t x = (f ^ mask1) - mask1, y = (u ^ mask1) - mask1, z = (v ^ mask1) - mask1;     // conditionally negated f, u, v
        g += x & mask2; q += y & mask2; r += z & mask2;
        mask1 &= mask2;                                    // swap when zeta < 0 and g odd
        zeta = (int32_t)((uint32_t)zeta ^ mask1) - 1;
        f += g & mask1; u += q & mask1; v += r & mask1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    tu = (int32_t)u; tv = (int32_t)v; tq = (int32_t)q; tr = (int32_t)r;
    return zeta;
}
// [d, e] <- t * [d, e] / 2^30  (mod q), exact thanks to the multiple of q that zeroes the low 30 bits
BN_FN void divsteps_update_de(Signed30 &d, Signed30 &e, int32_t u, int32_t v, int32_t q, int32_t r) {
    const int32_t M30 = 0x3fffffff;
    int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int32_t di = d.v[0], ei = e.v[0];
    int64_t cd = (int64_t)u * di + (int64_t)v * ei, ce = (int64_t)q * di + (int64_t)r * ei;
    md -= (int32_t)((k::QINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((k::QINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)k::Q30[0] * md;
    ce += (int64_t)k::Q30[0] * me;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        di = d.v[i]; ei = e.v[i];
        cd += (int64_t)u * di + (int64_t)v * ei + (int64_t)k::Q30[i] * md;
        ce += (int64_t)q * di + (int64_t)r * ei + (int64_t)k::Q30[i] * me;
        d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
    }
    d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
}
// [f, g] <- t * [f, g] / 2^30 (exact)
BN_FN void divsteps_update_fg(Signed30 &f, Signed30 &g, int32_t u, int32_t v, int32_t q, int32_t r) {
    const int32_t M30 = 0x3fffffff;
    int32_t fi = f.v[0], gi = g.v[0];
    int64_t cf = (int64_t)u * fi + (int64_t)v * gi, cg = (int64_t)q * fi + (int64_t)r * gi;
    cf >>= 30; cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        fi = f.v[i]; gi = g.v[i];
        cf += (int64_t)u * fi + (int64_t)v * gi;
        cg += (int64_t)q * fi + (int64_t)r * gi;
        f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
    }
    f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
}
BN_FN Fe fe_inverse_body(const Fe &a_in) {
    BN_REQUIRE(a_in.lb <= 2 && a_in.vb <= 8 && !a_in.sg, "fe_inverse input");
    // exact residue of the (lazy, Montgomery-form) input as a 256-bit integer
    uint32_t w[8];
    fe_pack_u32x8(fe_canonical(fe_mul(a_in, fe_const(k::ONE))), w);
    Signed30 f, g, d, e;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)w[wi] | (wi + 1 < 8 ? ((uint64_t)w[wi + 1] << 32) : 0);
        g.v[i] = (int32_t)((uint32_t)(two >> sh) & 0x3fffffffu);
        f.v[i] = k::Q30[i];
        d.v[i] = 0;
        e.v[i] = i == 0 ? 1 : 0;
    }
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; ++it) {
        int32_t tu, tv, tq, tr;
        zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], tu, tv, tq, tr);
        divsteps_update_de(d, e, tu, tv, tq, tr);
        divsteps_update_fg(f, g, tu, tv, tq, tr);
    }
    // g = 0, f = +-1 (or f = +-q when the input was 0, in which case d = 0): result = sign(f) * d, brought into [0, q)
    const int32_t M30 = 0x3fffffff;
    int32_t r[9];
    int32_t cond_add = d.v[8] >> 31, cond_neg = f.v[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) { int32_t t = d.v[i] + (k::Q30[i] & cond_add); r[i] = (t ^ cond_neg) - cond_neg; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { r[i + 1] += r[i] >> 30; r[i] &= M30; }
    cond_add = r[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] += k::Q30[i] & cond_add;
#pragma unroll
    for (int i = 0; i < 8; ++i) { r[i + 1] += r[i] >> 30; r[i] &= M30; }
    // 9 x 30-bit -> 8 x 32-bit words -> 9 x 29-bit limbs
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int bit = 32 * j, li = bit / 30, sh = bit - 30 * li;
        uint64_t vv = (uint64_t)(uint32_t)r[li] >> sh;
        if (li + 1 < 9) vv |= (uint64_t)(uint32_t)r[li + 1] << (30 - sh);
        w[j] = (uint32_t)vv;
    }
    // (x R)^-1 = x^-1 R^-1  ->  x^-1 R
    return fe_mul(fe_unpack_u32x8(w), fe_const(k::C_R3));
}
BN_LEAF1(fe_inverse, fe_inverse_body)

}  // namespace bn254

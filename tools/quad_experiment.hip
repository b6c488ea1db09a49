// Experiment (round-2 verdict item 6b, "four lanes per pairing"): the two operations the final exponentiation is made of - the
// Granger-Scott cyclotomic squaring and the Fq12 product - with ONE Fq12 spread over FOUR lanes instead of two, against the
// lane-pair mapping of the shipped kernels (bn_amd/csrc/tower.hpp), on the chain shape of an exponentiation by u (runs of five
// squarings, then a product).  Quad mapping: the lower lane pair of a quad holds c0 (an Fq6 = three Fq2, each still even lane c0 /
// odd lane c1), the upper pair holds c1; the pairs talk through DPP quad_perm [2,3,0,1].
//   product: lower computes a0 b0, upper a1 b1 (six Fq2 products each, at once); the Karatsuba cross term (a0+a1)(b0+b1) is six
//            more - three per pair; 9 Fq2 products per pair instead of 18.
//   squaring: each of the three Fp4 squarings takes the product a b and the product (a + b)(a + xi b): one per pair; 3 instead of 6.
// Both kernels store canonical bytes; the host compares them (the two mappings must agree bit for bit) and times them at element
// counts where the lane-pair mapping leaves SIMDs empty (the regime the question is about) and where it fills the chip.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bn_amd/csrc tools/quad_experiment.hip -o build_variants/quad_experiment
#define BN_INLINE_ALL 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "tower.hpp"
#include "io.hpp"

using namespace bn254;
typedef Fq2B<Fe> F2;

__device__ __forceinline__ Fe xq(const Fe &x) {            // the same limb of the OTHER lane pair of the quad
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[i], 0x4E, 0xF, 0xF, true);
    return r;
}
__device__ __forceinline__ F2 xq(const F2 &a) { return {xq(a.v)}; }
__device__ __forceinline__ Fq6<F2> xq(const Fq6<F2> &a) { return {xq(a.c0), xq(a.c1), xq(a.c2)}; }
__device__ __forceinline__ Fq6<F2> sel6(bool up, const Fq6<F2> &lo, const Fq6<F2> &hi) {
    return {f2_select(up, lo.c0, hi.c0), f2_select(up, lo.c1, hi.c1), f2_select(up, lo.c2, hi.c2)};
}

// my half (lower pair: c0, upper pair: c1) of a * b   (fq12.rs:295-307 with Karatsuba over Fq6, as tower.hpp f12_mul_src)
__device__ __forceinline__ Fq6<F2> quad_f12_mul(const Fq6<F2> &a, const Fq6<F2> &b, bool up) {
    const Fq6<F2> oa = xq(a), ob = xq(b);
    const Fq6<F2> mine = f6_mul(a, b);                                       // lower: aa = a0 b0, upper: bb = a1 b1
    const Fq6<F2> s = f6_add_norm(a, oa), t = f6_add_norm(b, ob);            // a0 + a1, b0 + b1 on both pairs
    // the cross product s * t (tower.hpp f6_mul): lower takes s0 t0, s1 t1, s2 t2, upper the three Karatsuba sums
    const F2 x0 = f2_select(up, s.c0, f2_add(s.c1, s.c2)), y0 = f2_select(up, t.c0, f2_norm(f2_add(t.c1, t.c2)));
    const F2 x1 = f2_select(up, s.c1, f2_add(s.c0, s.c1)), y1 = f2_select(up, t.c1, f2_norm(f2_add(t.c0, t.c1)));
    const F2 x2 = f2_select(up, s.c2, f2_add(s.c0, s.c2)), y2 = f2_select(up, t.c2, f2_norm(f2_add(t.c0, t.c2)));
    const F2 p0 = f2_mul(x0, y0), p1 = f2_mul(x1, y1), p2 = f2_mul(x2, y2);
    const F2 q0 = xq(p0), q1 = xq(p1), q2 = xq(p2);
    const F2 v0 = f2_select(up, p0, q0), v1 = f2_select(up, p1, q1), v2 = f2_select(up, p2, q2);            // s_i t_i
    const F2 k12 = f2_select(up, q0, p0), k01 = f2_select(up, q1, p1), k02 = f2_select(up, q2, p2);         // the sums' products
    const Fq6<F2> other = xq(mine);
    const Fq6<F2> aa = sel6(up, mine, other), bb = sel6(up, other, mine);
    // lower: c0 = aa + v bb;  upper: c1 = s t - aa - bb
    Fq6<F2> lo, hi;
    lo.c0 = f2_lc_xi<1, 1>(bb.c2, aa.c0);
    lo.c1 = f2_lc3<1, 1, 0>(aa.c1, bb.c0, bb.c0);
    lo.c2 = f2_lc3<1, 1, 0>(aa.c2, bb.c1, bb.c1);
    Fq6<F2> st;
    st.c0 = f2_lc_xi<1, 1>(f2_ssub(f2_ssub(k12, v1), v2), v0);
    st.c1 = f2_lc_xi<1, 1>(v2, f2_ssub(f2_ssub(k01, v0), v1));
    st.c2 = f2_lc3<1, -1, -1>(f2_add(k02, v1), v0, v2);
    hi = f6_lc3<1, -1, -1>(st, aa, bb);
    return sel6(up, lo, hi);
}

// my half of the Granger-Scott squaring (fq12.rs:178-227 as tower.hpp f12_cyclotomic_sqr)
__device__ __forceinline__ Fq6<F2> quad_cyclotomic_sqr(const Fq6<F2> &h, bool up) {
    const Fq6<F2> o = xq(h);
    // lower holds (z0, z4, z3) = c0, upper (z2, z1, z5) = c1
    const F2 z0 = f2_select(up, h.c0, o.c0), z4 = f2_select(up, h.c1, o.c1), z3 = f2_select(up, h.c2, o.c2);
    const F2 z2 = f2_select(up, o.c0, h.c0), z1 = f2_select(up, o.c1, h.c1), z5 = f2_select(up, o.c2, h.c2);
    auto fp4 = [&](const F2 &a, const F2 &b, F2 &tmp, F2 &m) {             // lower: tmp = a b, upper: m = (a + b)(a + xi b)
        const F2 x = f2_select(up, a, f2_add(a, b)), y = f2_select(up, b, f2_lc_xi<1, 1>(b, a));
        const F2 p = f2_mul(x, y), q = xq(p);
        tmp = f2_select(up, p, q); m = f2_select(up, q, p);
    };
    F2 t01, m01, t23, m23, t45, m45;
    fp4(z0, z1, t01, m01); fp4(z2, z3, t23, m23); fp4(z4, z5, t45, m45);
    Fq6<F2> lo, hi;
    lo.c0 = f2_lc_xi2<-3, 3, -2>(t01, f2_ssub(m01, t01), z0);
    lo.c1 = f2_lc_xi2<-3, 3, -2>(t23, f2_ssub(m23, t23), z4);
    lo.c2 = f2_lc_xi2<-3, 3, -2>(t45, f2_ssub(m45, t45), z3);
    hi.c0 = f2_lc_xi<6, 2>(t45, z2);
    hi.c1 = f2_lc3<6, 2, 0>(t01, z1, z1);
    hi.c2 = f2_lc3<6, 2, 0>(t23, z5, z5);
    return sel6(up, lo, hi);
}

// the multiplier b waits in LDS, [dword][lane] (the shipped final exponentiation keeps its operands in a table too: an Fq12 held
// in registers next to the running value does not fit 256 VGPRs)
struct LdsFq6 {
    uint32_t *base;          // this lane's column
    __device__ __forceinline__ void st(int k, const Fq6<F2> &v) const {
#pragma unroll
        for (int i = 0; i < 9; ++i) { base[((k * 27) + i) * 64] = v.c0.v.l[i]; base[((k * 27) + 9 + i) * 64] = v.c1.v.l[i]; base[((k * 27) + 18 + i) * 64] = v.c2.v.l[i]; }
    }
    __device__ __forceinline__ Fq6<F2> ld(int k) const {
        Fq6<F2> v;
#pragma unroll
        for (int i = 0; i < 9; ++i) { v.c0.v.l[i] = base[((k * 27) + i) * 64]; v.c1.v.l[i] = base[((k * 27) + 9 + i) * 64]; v.c2.v.l[i] = base[((k * 27) + 18 + i) * 64]; }
        return v;
    }
};
struct LdsFq12Src {
    const LdsFq6 &t;
    __device__ __forceinline__ Fq6<F2> c0() const { return t.ld(0); }
    __device__ __forceinline__ Fq6<F2> c1() const { return t.ld(1); }
};

// chain: `rounds` x (five squarings, one product by b)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
pair_chain(const uint32_t *a_in, const uint32_t *b_in, uint32_t *out, uint32_t n, int rounds) {
    uint32_t e = (blockIdx.x * 64 + threadIdx.x) >> 1;
    const bool live = e < n;
    if (!live) e = n - 1;
    __shared__ uint32_t park[54 * 64];
    const LdsFq6 tb = {park + threadIdx.x};
    Fq12<F2> a = f12_load<F2>(a_in + 96u * e);
    {
        const Fq12<F2> b = f12_load<F2>(b_in + 96u * e);
        tb.st(0, b.c0); tb.st(1, b.c1);
    }
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
#pragma unroll 1
        for (int s = 0; s < 5; ++s) a = f12_cyclotomic_sqr(a);
        a = f12_mul_src(a, LdsFq12Src{tb}, false);
    }
    if (live) f12_store(a, out + 96u * e);
}
__device__ __forceinline__ Fq6<F2> load_half(const uint32_t *w) {
    return {f2_load((const F2 *)nullptr, w), f2_load((const F2 *)nullptr, w + 16), f2_load((const F2 *)nullptr, w + 32)};
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
quad_chain(const uint32_t *a_in, const uint32_t *b_in, uint32_t *out, uint32_t n, int rounds) {
    uint32_t e = (blockIdx.x * 64 + threadIdx.x) >> 2;
    const bool live = e < n, up = (threadIdx.x & 2u) != 0;
    if (!live) e = n - 1;
    __shared__ uint32_t park[27 * 64];
    const LdsFq6 tb = {park + threadIdx.x};
    Fq6<F2> a = load_half(a_in + 96u * e + (up ? 48u : 0u));
    tb.st(0, load_half(b_in + 96u * e + (up ? 48u : 0u)));
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
#pragma unroll 1
        for (int s = 0; s < 5; ++s) a = quad_cyclotomic_sqr(a, up);
        a = quad_f12_mul(a, tb.ld(0), up);
    }
    if (live) {
        uint32_t *w = out + 96u * e + (up ? 48u : 0u);
        f2_store(a.c0, w); f2_store(a.c1, w + 16); f2_store(a.c2, w + 32);
    }
}

static uint64_t sm(uint64_t &s) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

int main() {
    const uint32_t nmax = 1u << 16;
    const int rounds = 12;                               // 60 squarings + 12 products: one exponentiation by u, roughly
    std::vector<uint64_t> ha((size_t)nmax * 48), hb((size_t)nmax * 48);
    uint64_t seed = 7;
    for (size_t i = 0; i < ha.size(); ++i) {             // field elements below q: top limb below 0x30644e72e131a029
        ha[i] = sm(seed); hb[i] = sm(seed);
        if (i % 4 == 3) { ha[i] %= 0x30644e72e131a029ull; hb[i] %= 0x30644e72e131a029ull; }
    }
    uint32_t *da, *db, *o1, *o2;
    const size_t bytes = (size_t)nmax * 384;
    hipMalloc(&da, bytes); hipMalloc(&db, bytes); hipMalloc(&o1, bytes); hipMalloc(&o2, bytes);
    hipMemcpy(da, ha.data(), bytes, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), bytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("chain: %d x (5 cyclotomic squarings + 1 product) per element; ms per launch, best of 3\n", rounds);
    printf("%8s %12s %12s %8s   %s\n", "elements", "lane pair", "four lanes", "ratio", "waves per SIMD (pair / quad) on 256 CUs");
    bool all_equal = true;
    for (uint32_t n : {256u, 2048u, 8192u, 16384u, 32768u, 65536u}) {
        float best[2] = {1e9f, 1e9f};
        for (int rep = 0; rep < 4; ++rep) {
            float ms;
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(pair_chain, dim3((n * 2 + 63) / 64), dim3(64), 0, 0, da, db, o1, n, rounds);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best[0]) best[0] = ms;
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(quad_chain, dim3((n * 4 + 63) / 64), dim3(64), 0, 0, da, db, o2, n, rounds);
            hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best[1]) best[1] = ms;
        }
        std::vector<uint64_t> r1((size_t)n * 48), r2((size_t)n * 48);
        hipMemcpy(r1.data(), o1, (size_t)n * 384, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, (size_t)n * 384, hipMemcpyDeviceToHost);
        const bool eq = memcmp(r1.data(), r2.data(), (size_t)n * 384) == 0;
        all_equal = all_equal && eq;
        printf("%8u %12.3f %12.3f %8.2f   %.2f / %.2f   %s\n", n, best[0], best[1], best[0] / best[1], n * 2 / 64.0 / 1024.0, n * 4 / 64.0 / 1024.0, eq ? "equal" : "DIFFERENT");
    }
    printf("%s\n", all_equal ? "all results bit-identical between the two mappings" : "MISMATCH");
    return all_equal ? 0 : 1;
}

// Wave-cooperative kernels of the BN254 engine for MI355X (gfx950): ONE Fq12 per WAVE (wave.hpp) instead of one per lane pair.
// For the latency-bound tails of the path: the single final exponentiation of a multi-pairing (fq12.rs:41-88 behind the fold of
// shootout/main.rs:11-16), the combination of the per-GPU partial products, and batches too small to fill the chip - where the
// lane-pair kernels of bn254_kernels_b.hip take 2.1 ms per final exponentiation whatever the batch size.
//
// One workgroup = one wave = one Fq12.  LDS: the register file of the machine (5 pages x 9 limbs x 64 slots = 11.5 KB), nothing
// else - the role tables (55 phases x 32 pairs x 24 B = 42 KB, the same for every wave) stay in global memory: L2 hits, and the
// interpreter asks for the next phase's role while this phase computes.  A copy per workgroup in LDS was no faster for one pairing
// (1.045 against 1.052 ms) and held a CU to three workgroups, one wave per SIMD - and a lone wave leaves half of the multiplier idle
// (it issues a v_mad_u64_u32 every ~8 cycles whatever the dependencies: profiles/r03s_ubench_chain.txt).  With 11.5 KB a CU takes
// thirteen: 1024 pairings cost what one does, 2048 1.5 ms instead of 3.2 ms (profiles/r03s_wave_roles_ab.txt).
#define BN_INLINE_ALL 1       // fe.hpp: leaves and Fq6/Fq12-sized steps force-inlined
#include <hip/hip_runtime.h>
#include "wave.hpp"
#include "io.hpp"

using namespace bn254;
using namespace bn254::wv;

namespace {
constexpr int ROLE_DWORDS_PER_PHASE = 32 * (int)(sizeof(Role) / 4);

struct WaveDev {
    using T = Fe;
    char *regs;              // register file in LDS, already offset to this lane's component (even lane c0, odd lane c1)
    __device__ __forceinline__ Fe ld(uint32_t off) const {
        Fe v;
#pragma unroll
        for (int i = 0; i < 9; ++i) v.l[i] = *(const uint32_t *)(regs + off + 256u * i);
        return v;
    }
    __device__ __forceinline__ void st(uint32_t off, const Fe &v) const {
#pragma unroll
        for (int i = 0; i < 9; ++i) *(uint32_t *)(regs + off + 256u * i) = v.l[i];
    }
    __device__ __forceinline__ Role role(uint32_t phase) const { return ((const Role *)&ROLES[0][0])[phase * 32u + (threadIdx.x >> 1)]; }
    __device__ __forceinline__ int pair() const { return (int)(threadIdx.x >> 1); }
    __device__ __forceinline__ void sync() const { __syncthreads(); }        // single-wave workgroup: a wave-level barrier
};
struct WaveLds { uint32_t regs[NPAGES * PAGE_DW]; };

// register file zeroed, Frobenius multipliers (and the Miller program's constants) into their registers
__device__ __forceinline__ WaveDev wave_init(WaveLds &l) {
    for (int i = threadIdx.x; i < NPAGES * PAGE_DW; i += 64) l.regs[i] = 0;
    __syncthreads();
    WaveDev w = {(char *)l.regs + 4u * (threadIdx.x & 1u)};
    const uint32_t j = threadIdx.x >> 1;
    if (j < 18) {
        Fe c;
#pragma unroll
        for (int i = 0; i < 9; ++i) c.l[i] = KCONST[j][threadIdx.x & 1u][i];
        w.st((uint32_t)KBASE_OFF[1] + 8u * j, c);
    }
    if (j >= 18 && j < 18 + NMCONST) {                      // constants of the Miller program
        Fe c;
#pragma unroll
        for (int i = 0; i < 9; ++i) c.l[i] = MCONST[j - 18][threadIdx.x & 1u][i];
        w.st(MCONST_OFF[j - 18], c);
    }
    return w;
}

// out[b] = final_exponentiation(f_in[b])   (fq12.rs:86-88), one workgroup per element
__global__ void __launch_bounds__(64) bn254_final_exp_W(const uint32_t *f_in, uint32_t *out) {
    __shared__ WaveLds lds;
    WaveDev w = wave_init(lds);
    w_load_f12(w, f_in + 96u * blockIdx.x, OFF_RES);
    w.sync();
    w_run(w, PROG_FE);
    w_store_f12(w, OFF_RES, out + 96u * blockIdx.x);
}

// out[b] = pairing(p[b], q[b])  (lib.rs:181-183 -> groups/mod.rs:764-771), or only its Miller value (final_exp = 0): the whole pairing
// as ONE program of the wave machine - prologue (both affine conversions behind one inversion phase), the fused NAF Miller loop on
// the isomorphic curve (six phases per doubling step, seven per addition step), the final exponentiation.  ~1.2 ms for one pairing
// where the lane-pair kernels need 2.3 + 0.6 ms; one workgroup per pairing, for batches that cannot fill the chip anyway.
__global__ void __launch_bounds__(64) bn254_pairing_W(const uint32_t *g1, const uint32_t *g2, uint32_t *out, int final_exp) {
    __shared__ WaveLds lds;
    WaveDev w = wave_init(lds);
    const uint32_t *w1 = g1 + 24u * blockIdx.x, *w2 = g2 + 48u * blockIdx.x;
    uint32_t zp = 0, zq = 0;
    for (int i = 0; i < 8; ++i) zp |= w1[16 + i];
    for (int i = 0; i < 16; ++i) zq |= w2[32 + i];
    const bool inf = zp == 0 || zq == 0;                                       // groups/mod.rs:766
    w_load_points(w, w1, w2);
    w.sync();
    w_run(w, final_exp ? PROG_PAIRING : PROG_MILLER);
    if (inf) { w_set_one(w); w.sync(); }
    w_store_f12(w, OFF_RES, out + 96u * blockIdx.x);
}

// out[b] = in[b*m] * in[b*m + 1] * ... * in[b*m + m-1], then optionally its final exponentiation: the tail of a multi-pairing
// (the per-GPU partial products of bn254_pairing_product_multi, or the survivors of the product tree), one workgroup per group
__global__ void __launch_bounds__(64) bn254_gt_tail_W(const uint32_t *in, uint32_t m, uint32_t *out, int final_exp) {
    __shared__ WaveLds lds;
    WaveDev w = wave_init(lds);
    const uint32_t *src = in + 96u * m * blockIdx.x;
    w_load_f12(w, src, OFF_RES);
    w.sync();
#pragma unroll 1
    for (uint32_t j = 1; j < m; ++j) {
        w_load_f12(w, src + 96u * j, OFF_SLOT0);
        w.sync();
        w_run(w, PROG_MUL);
    }
    if (final_exp) w_run(w, PROG_FE);
    w_store_f12(w, OFF_RES, out + 96u * blockIdx.x);
}

// measurement: `iters` runs of one program on one wave (which: 0 cyclotomic squaring = 2 phases, 1 product = 3 phases, 2 slot copy =
// 1 COMB phase, 3 Frobenius map = 1 PROD phase with conjugation, 4 the whole final exponentiation, 5 a fused run of five squarings)
__global__ void __launch_bounds__(64) bn254_wave_ubench_W(int which, int iters, uint32_t *out) {
    __shared__ WaveLds lds;
    WaveDev w = wave_init(lds);
    w.sync();
    if (threadIdx.x < 12) w.st(OFF_RES + 8u * (threadIdx.x >> 1), w.ld((uint32_t)KBASE_OFF[1] + 8u * (threadIdx.x >> 1) + 8u));
    w.sync();
    const uint32_t *prog = which == 0 ? PROG_CYC : which == 1 ? PROG_MUL : which == 2 ? PROG_PUT0 : which == 3 ? PROG_FROB1 : which == 4 ? PROG_FE : PROG_CYC5;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) w_run(w, prog);
    w_store_f12(w, OFF_RES, out);
}

// ---- the multi-pairing product tree in ONE launch (lane chunks -> wave -> grid, last arriver continues) ------------
// in[0..n) Fq12 values (384-byte images) -> out[0] = their product (fq12.rs:295-307 folded, shootout/main.rs:11-16; the order is
// free: Fq12 is commutative and every value is exact).
//   (a) lane pair g of the grid (the first `per_wave` pairs of each wave) multiplies the `chunk` consecutive values of its group in
//       the lane-pair mapping (full-width work);
//   (b) the partial products of a wave: `bfly` butterfly levels of lane-pair products (all pairs at once, each level halves what
//       is left), then the wave-cooperative product folds the rest one after the other (~2.7 us each instead of ~20 us);
//   (c) waves meet pairwise in a binary tree over the wave index: each arrival publishes its value and takes a ticket on the node;
//       the FIRST arriver exits, the SECOND one multiplies the two values and moves up - nobody ever waits, so the grid may be
//       larger than what is resident.  Publication is plain stores -> agent-scope release -> ticket; consumption is ticket ->
//       agent-scope acquire -> plain loads (cdna_hip_programming.md G16: per-XCD L2s are not coherent with each other).
// `counters` (one u32 per tree node, < gridDim.x of them) must be zero at launch; `scratch` holds 108 dwords per node.
typedef Fq2B<Fe> F2;
struct WaveDevR {                                   // the machine on a register file of two pages (the product's operands and temporaries)
    using T = Fe;
    char *regs;
    __device__ __forceinline__ Fe ld(uint32_t off) const {
        Fe v;
#pragma unroll
        for (int i = 0; i < 9; ++i) v.l[i] = *(const uint32_t *)(regs + off + 256u * i);
        return v;
    }
    __device__ __forceinline__ void st(uint32_t off, const Fe &v) const {
#pragma unroll
        for (int i = 0; i < 9; ++i) *(uint32_t *)(regs + off + 256u * i) = v.l[i];
    }
    __device__ __forceinline__ Role role(uint32_t phase) const { return ((const Role *)&ROLES[0][0])[phase * 32u + (threadIdx.x >> 1)]; }
    __device__ __forceinline__ int pair() const { return (int)(threadIdx.x >> 1); }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
};
constexpr uint32_t OFF_OPND = (uint32_t)KBASE_OFF[1];          // operand registers of PROG_MULR
constexpr int NODE_DWORDS = 108;                               // 6 Fq2 x 2 components x 9 limbs
__device__ __forceinline__ void put_f12(const WaveDevR &w, uint32_t reg0, const Fq12<F2> &f) {
    w.st(reg0, f.c0.c0.v); w.st(reg0 + 8, f.c0.c1.v); w.st(reg0 + 16, f.c0.c2.v);
    w.st(reg0 + 24, f.c1.c0.v); w.st(reg0 + 32, f.c1.c1.v); w.st(reg0 + 40, f.c1.c2.v);
}
__device__ __forceinline__ Fe xchg_fe(const Fe &a, int lane_xor) {
    Fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)__shfl_xor((int)a.l[i], lane_xor, 64);
    return r;
}
// the same value as held by the lane pair `lane_xor / 2` pairs away
__device__ __forceinline__ Fq12<F2> xchg_f12(const Fq12<F2> &f, int lane_xor) {
    Fq12<F2> r;
    r.c0.c0.v = xchg_fe(f.c0.c0.v, lane_xor); r.c0.c1.v = xchg_fe(f.c0.c1.v, lane_xor); r.c0.c2.v = xchg_fe(f.c0.c2.v, lane_xor);
    r.c1.c0.v = xchg_fe(f.c1.c0.v, lane_xor); r.c1.c1.v = xchg_fe(f.c1.c1.v, lane_xor); r.c1.c2.v = xchg_fe(f.c1.c2.v, lane_xor);
    return r;
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
bn254_gt_reduce_W(const uint32_t *in, uint32_t n, uint32_t chunk, uint32_t per_wave, uint32_t bfly, uint32_t *scratch, uint32_t *counters, uint32_t *out) {
    __shared__ struct { uint32_t regs[2 * PAGE_DW]; } lds;
    for (int i = threadIdx.x; i < 2 * PAGE_DW; i += 64) lds.regs[i] = 0;
    __syncthreads();
    WaveDevR w = {(char *)lds.regs + 4u * (threadIdx.x & 1u)};
    const uint32_t pair = threadIdx.x >> 1;
    // (a) this lane pair's group
    const uint32_t groups = (n + chunk - 1) / chunk, g = blockIdx.x * per_wave + pair;
    const bool live = pair < per_wave && g < groups;
    const uint32_t lo = live ? g * chunk : 0u, hi = live ? (lo + chunk < n ? lo + chunk : n) : 1u;
    Fq12<F2> acc = f12_load<F2>(in + 96u * lo);
#pragma unroll 1
    for (uint32_t j = lo + 1; j < hi; ++j) acc = f12_mul_o(acc, f12_load<F2>(in + 96u * j));
    // (b) fold the wave's live partial products: `bfly` butterfly levels of lane-pair products (all pairs at once: pair p takes
    // the value of pair p ^ d), then the wave-cooperative product over what is left (every 2^bfly-th pair)
    const uint32_t live_pairs = groups - blockIdx.x * per_wave < per_wave ? groups - blockIdx.x * per_wave : per_wave;      // >= 1 by the grid size
    uint32_t stride = 1;
    if (bfly && live_pairs > 1) {
        if (!live) acc = f12_one<F2>();
#pragma unroll 1
        for (; stride < (1u << bfly) && stride < live_pairs; stride <<= 1) acc = f12_mul_o(acc, xchg_f12(acc, 2 * stride));
    }
    if (pair == 0) put_f12(w, OFF_RES, acc);
#pragma unroll 1
    for (uint32_t j = stride; j < live_pairs; j += stride) {
        if (pair == j) put_f12(w, OFF_OPND, acc);
        w.sync();
        w_run(w, PROG_MULR);
    }
    w.sync();
    // (c) arrival tree over the wave index
    uint32_t idx = blockIdx.x, cnt = gridDim.x, coff = 0, soff = 0;
#pragma unroll 1
    while (cnt > 1) {
        if ((idx ^ 1u) < cnt) {
            uint32_t *mine = scratch + (size_t)(soff + idx) * NODE_DWORDS, *theirs = scratch + (size_t)(soff + (idx ^ 1u)) * NODE_DWORDS;
            if (pair < 6) {
                const Fe v = w.ld(OFF_RES + 8u * pair);
#pragma unroll
                for (int i = 0; i < 9; ++i) mine[threadIdx.x * 9 + i] = v.l[i];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint32_t ticket = 0;
            if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(counters + coff + (idx >> 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
            if (ticket == 0) return;                           // first at this node: the sibling's wave carries on with both values
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
            if (pair < 6) {
                Fe v;
#pragma unroll
                for (int i = 0; i < 9; ++i) v.l[i] = theirs[threadIdx.x * 9 + i];
                w.st(OFF_OPND + 8u * pair, v);
            }
            w.sync();
            w_run(w, PROG_MULR);
        }
        soff += cnt; coff += (cnt + 1) >> 1; idx >>= 1; cnt = (cnt + 1) >> 1;
    }
    w_store_f12(w, OFF_RES, out);
}
}  // namespace

extern "C" {
// scratch bytes and counter words the product tree needs for `n` values in groups of `chunk`
void bn254_gt_reduce_sizes_W(size_t n, unsigned chunk, unsigned per_wave, size_t *grid, size_t *scratch_bytes, size_t *counter_words) {
    const size_t groups = (n + chunk - 1) / chunk, waves = (groups + per_wave - 1) / per_wave;
    // tree levels have ceil(c/2) nodes each: sum over levels of the values published <= 2 waves + log2(waves), tickets <= waves + log2(waves)
    *grid = waves; *scratch_bytes = (2 * waves + 64) * NODE_DWORDS * sizeof(uint32_t); *counter_words = waves + 64;
}
int bn254_launch_gt_reduce_W(const void *in, size_t n, unsigned chunk, unsigned per_wave, unsigned bfly, void *scratch, void *counters, void *out, hipStream_t s) {
    size_t grid, sb, cw;
    bn254_gt_reduce_sizes_W(n, chunk, per_wave, &grid, &sb, &cw);
    hipError_t e = hipMemsetAsync(counters, 0, cw * sizeof(uint32_t), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bn254_gt_reduce_W, dim3((unsigned)grid), dim3(64), 0, s, (const uint32_t *)in, (uint32_t)n, chunk, per_wave, bfly, (uint32_t *)scratch, (uint32_t *)counters, (uint32_t *)out);
    return (int)hipGetLastError();
}
int bn254_launch_wave_ubench_W(int which, int iters, void *out, hipStream_t s) {
    hipLaunchKernelGGL(bn254_wave_ubench_W, dim3(1), dim3(64), 0, s, which, iters, (uint32_t *)out);
    return (int)hipGetLastError();
}
int bn254_launch_final_exp_W(const void *f, void *out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(bn254_final_exp_W, dim3((unsigned)n), dim3(64), 0, s, (const uint32_t *)f, (uint32_t *)out);
    return (int)hipGetLastError();
}
int bn254_launch_pairing_W(const void *p, const void *q, void *out, size_t n, int final_exp, hipStream_t s) {
    hipLaunchKernelGGL(bn254_pairing_W, dim3((unsigned)n), dim3(64), 0, s, (const uint32_t *)p, (const uint32_t *)q, (uint32_t *)out, final_exp);
    return (int)hipGetLastError();
}
int bn254_launch_gt_tail_W(const void *in, size_t groups, unsigned m, void *out, int final_exp, hipStream_t s) {
    hipLaunchKernelGGL(bn254_gt_tail_W, dim3((unsigned)groups), dim3(64), 0, s, (const uint32_t *)in, m, (uint32_t *)out, final_exp);
    return (int)hipGetLastError();
}
}

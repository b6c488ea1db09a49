// MI355X (gfx950) kernels and the C ABI of the batched BN254 pairing engine (include/bn254_hip.h).
// Built by bn_amd/_native.py build(): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-dpp-combine=false -c <each unit>, then one
// -shared link.  The -mllvm flag is REQUIRED in every unit (fe.hpp fe_lc4_core: a DPP fold into a reversed subtraction computes the wrong value on gfx950).
//
// This unit holds the host side (contexts, options, launch policy: which of the wave / four-lane / lane-pair kernels of the other units runs a
// call) and the wire-format kernels (one record per lane).  The pairing and scalar-multiplication kernels live in bn254_kernels_{b,q,w,mul}.hip.
// (The one-lane-per-pairing kernels of rounds 1-4 - "mapping A", a test double - moved to tests/testdouble/ in round 5.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bn254_hip.h"
#include "curve.hpp"
#include "io.hpp"
#include "io_wire.hpp"

using namespace bn254;

// ======================================================================================================== kernels
namespace {

constexpr int BLOCK = 64;

// wire format (io_wire.hpp): one record per lane; byte-granular global accesses (65/129-byte strides), not a hot path
__global__ void __launch_bounds__(BLOCK) bn254_g1_encode_k(const uint32_t *p, uint8_t *out, uint32_t n) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    uint32_t w[24];
    for (int i = 0; i < 24; ++i) w[i] = p[24u * idx + i];
    g1_encode_record(w, out + 65u * idx);
}
__global__ void __launch_bounds__(BLOCK) bn254_g2_encode_k(const uint32_t *p, uint8_t *out, uint32_t n) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    uint32_t w[48];
    for (int i = 0; i < 48; ++i) w[i] = p[48u * idx + i];
    g2_encode_record(w, out + 129u * idx);
}
__global__ void __launch_bounds__(BLOCK) bn254_g1_decode_k(const uint8_t *in, uint32_t *out, int32_t *status, uint32_t n) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    uint8_t b[65];
    for (int i = 0; i < 65; ++i) b[i] = in[65u * idx + i];
    status[idx] = g1_decode_record(b, out + 24u * idx);
}
__global__ void __launch_bounds__(BLOCK) bn254_g2_decode_k(const uint8_t *in, uint32_t *out, int32_t *status, uint32_t n) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    uint8_t b[129];
    for (int i = 0; i < 129; ++i) b[i] = in[129u * idx + i];
    status[idx] = g2_decode_record(b, out + 48u * idx);
}

__global__ void __launch_bounds__(BLOCK) bn254_fr_encode_k(const uint32_t *k, uint8_t *out, uint32_t n) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    uint32_t w[8];
    for (int i = 0; i < 8; ++i) w[i] = k[8u * idx + i];
    fr_encode_record(w, out + 32u * idx);
}
__global__ void __launch_bounds__(BLOCK) bn254_fr_decode_k(const uint8_t *in, uint32_t *out, int32_t *status, uint32_t n) {
    uint32_t idx = blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    uint8_t b[32];
    for (int i = 0; i < 32; ++i) b[i] = in[32u * idx + i];
    status[idx] = fr_decode_record(b, out + 8u * idx);
}

}  // namespace

// ======================================================================================================== host side
#include "host_ctx.hpp"

// bn254_multi.hip: chunked, double-buffered host-buffer path (pinned staging, one stream + worker thread per chunk in flight)
struct BnMapSpec;
int bn_pairing_batch_pipelined(bn254_ctx *ctx, const bn_g1 *p, const bn_g2 *q, bn_gt *out, size_t n);
int bn_mul_batch_pipelined(bn254_ctx *ctx, int g, const void *p, const bn_fr *k, void *out, size_t n);

namespace {

constexpr int MAX_DEFAULT_CTX = 64;
std::mutex g_default_mu;
bn254_ctx *g_default[MAX_DEFAULT_CTX] = {};

inline unsigned grid_for(size_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

// folds the oldest finished records into the per-name totals and recycles their events (bounded memory with profiling left on)
void fold_records(bn254_ctx *c, size_t keep) {
    size_t drop = c->recs.size() > keep ? c->recs.size() - keep : 0;
    for (size_t i = 0; i < drop; ++i) {
        auto &r = c->recs[i];
        float ms = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto &t = c->folded[r.name];
            t.first += ms; t.second += 1;
        }
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    c->recs.erase(c->recs.begin(), c->recs.begin() + drop);
}

}  // namespace

BnScope::BnScope(bn254_ctx *c_, hipStream_t s_, const char *n) : c(c_), s(s_), on(c_->profile), name(n) {
    if (on) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, s); }
}
BnScope::~BnScope() {
    if (!on) return;
    hipEventRecord(b, s);
    std::lock_guard<std::mutex> lk(c->prof_mu);
    c->recs.push_back({name, a, b});
    if (c->recs.size() > 4096) fold_records(c, 2048);
}

BnScratchGuard::BnScratchGuard(bn254_ctx *c_, hipStream_t s_) : c(c_), s(s_), rc(BN254_OK) {
    c->scratch_mu.lock();
    if (c->scratch_used && c->scratch_stream != s) rc = (int)hipStreamWaitEvent(s, c->scratch_ev, 0);
}
BnScratchGuard::~BnScratchGuard() {
    if (!c->scratch_ev) hipEventCreateWithFlags(&c->scratch_ev, hipEventDisableTiming);
    if (c->scratch_ev && hipEventRecord(c->scratch_ev, s) == hipSuccess) { c->scratch_used = true; c->scratch_stream = s; }
    c->scratch_mu.unlock();
}

int bn_get_ctx(bn254_ctx *&ctx) {
    if (ctx) return BN254_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEFAULT_CTX) return BN254_E_NO_DEVICE;
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default[dev]) {
        int rc = bn254_ctx_create(dev, &g_default[dev]);
        if (rc) return rc;
    }
    ctx = g_default[dev];
    return BN254_OK;
}

// ---- tunables (include/bn254_hip.h BN254_OPT_*).  Raw values live in the context, < 0 meaning "default"; the defaults are functions of
// the device's CU count, so a partition or a smaller part gets thresholds that fit it.
// The ONLY place this library reads its debug environment (BN254_RCCL_PATH, a file path, is read where RCCL is loaded): once per
// process, into the seed values every new context starts from.
constexpr size_t BN_LAUNCH_MAX = (size_t)1 << 22;       // units per launch (32-bit word offsets inside a kernel); also the cap of the size options
namespace {
bool bn_opt_valid(int key, long v);
struct DebugEnv { long opt[BN254_OPT_COUNT_]; int exchange; bool affinity; };
const DebugEnv &bn_debug_env() {
    static DebugEnv env;
    static std::once_flag once;
    std::call_once(once, [] {
        for (long &v : env.opt) v = -1;
        env.exchange = BN254_EXCHANGE_AUTO;
        env.affinity = true;
        static const struct { const char *name; int key; } vars[] = {
            {"BN254_WAVE_PAIRING_MAX", BN254_OPT_WAVE_PAIRING_MAX}, {"BN254_WAVE_FE_MAX", BN254_OPT_WAVE_FE_MAX}, {"BN254_QUAD_MAX", BN254_OPT_QUAD_MAX},
            {"BN254_MILLER_SHARED", BN254_OPT_MILLER_SHARED}, {"BN254_GT_POW_MODE", BN254_OPT_GT_POW_MODE}, {"BN254_PRODUCT_CHUNK", BN254_OPT_PRODUCT_CHUNK},
            {"BN254_PRODUCT_PER_WAVE", BN254_OPT_PRODUCT_PER_WAVE}, {"BN254_PRODUCT_BFLY", BN254_OPT_PRODUCT_BFLY}, {"BN254_ROUND_PAIRS", BN254_OPT_ROUND_PAIRS},
            {"BN254_PIPELINE_CHUNK", BN254_OPT_PIPELINE_CHUNK}, {"BN254_PIPELINE_SLOTS", BN254_OPT_PIPELINE_SLOTS},
            {"BN254_STREAM_STOP_AT_ERROR", BN254_OPT_STREAM_STOP_AT_ERROR}};
        for (const auto &v : vars)
            if (const char *e = getenv(v.name)) { const long x = atol(e); if (x >= 0 && bn_opt_valid(v.key, x)) env.opt[v.key] = x; }
        if (const char *e = getenv("BN254_MULTI_AFFINITY")) env.affinity = atoi(e) != 0;
        if (const char *e = getenv("BN254_MULTI_EXCHANGE")) env.exchange = !strcmp(e, "peer") ? BN254_EXCHANGE_PEER : !strcmp(e, "rccl") ? BN254_EXCHANGE_RCCL : BN254_EXCHANGE_AUTO;
    });
    return env;
}
// is `value` acceptable for `key`?  (negative values are always accepted: "restore the default")
// Sizes of ONE launch are capped at BN_LAUNCH_MAX = 2^22 units: the kernels index their inputs, outputs and context-owned tables with 32-bit
// word offsets (96 words per Fq12, 126 x lanes-in-the-launch table rows per lane), which 2^22 pairings keep below 2^32 in every mapping;
// larger batches are cut into sub-launches by the host whatever the thresholds say.
bool bn_opt_valid(int key, long v) {
    switch (key) {
        case BN254_OPT_WAVE_PAIRING_MAX: case BN254_OPT_WAVE_FE_MAX: case BN254_OPT_QUAD_MAX: return v <= (long)BN_LAUNCH_MAX;
        case BN254_OPT_MILLER_SHARED: return v == 0 || v == 1 || v == 2 || v == 4;
        case BN254_OPT_GT_POW_MODE: return v <= 2;
        case BN254_OPT_PRODUCT_CHUNK: return v >= 1 && v <= 4096;
        case BN254_OPT_PRODUCT_PER_WAVE: return v >= 1 && v <= 32;
        case BN254_OPT_PRODUCT_BFLY: return v <= 5;
        case BN254_OPT_ROUND_PAIRS: case BN254_OPT_PIPELINE_CHUNK: return v >= 1 && v <= (long)BN_LAUNCH_MAX;
        case BN254_OPT_PIPELINE_SLOTS: return v >= 1 && v <= BN_MAX_SLOTS;
        case BN254_OPT_STREAM_STOP_AT_ERROR: return v <= 1;
        default: return false;
    }
}
}  // namespace
int bn_debug_multi_exchange() { return bn_debug_env().exchange; }
bool bn_debug_multi_affinity() { return bn_debug_env().affinity; }
long bn_opt(const bn254_ctx *c, int key) {
    const long v = c->opt[key].load(std::memory_order_relaxed);
    if (v >= 0) return v;
    const long cus = c->cus > 0 ? c->cus : 256;
    switch (key) {
        // one pairing / exponentiation per WAVE: a workgroup needs 11.5 KB of LDS, thirteen fit a CU; a pairing takes 1.05 ms up to 4 per CU
        // (one wave per SIMD), 1.5 ms at 8, 2.05 ms at 12, 3.2 ms at 16 per CU - and 2.7 ms on four lanes (bn254_kernels_q.hip) whatever
        // the count up to 64 per CU; a final exponentiation 0.5 / 0.74 / 1.48 ms at 4 / 8 / 16 per CU against 1.24 ms on four lanes
        // (profiles/r04_wave_latency.json, 256 CUs).  They cross at ~14 and ~13 per CU.
        case BN254_OPT_WAVE_PAIRING_MAX: return (BN254_HAVE_QUAD ? 14 : 20) * cus;
        case BN254_OPT_WAVE_FE_MAX: return (BN254_HAVE_QUAD ? 13 : 20) * cus;
        case BN254_OPT_QUAD_MAX: return BN254_HAVE_QUAD ? 64 * cus : 0;
        case BN254_OPT_MILLER_SHARED: case BN254_OPT_GT_POW_MODE: case BN254_OPT_STREAM_STOP_AT_ERROR: return 0;
        case BN254_OPT_ROUND_PAIRS: return 256 * cus;
        case BN254_OPT_PIPELINE_SLOTS: return 2;
        default: return -1;                  // product shape, pipeline chunk: decided per call from the size
    }
}

// Launch granularity of the lane-pair kernels.  One "round" = 256 pairings per CU = two resident waves on every SIMD: the shape the
// s_setprio hand-over of bn254_kernels_b.hip is tuned for (inside a 2^18 launch every 2^16 cost 6 % more than alone, and a box with
// fewer than 256 CUs ran a fixed 2^16 launch as 1 + a fraction rounds: profiles/r02r_*).  Larger batches are issued as equal
// sub-launches of at most one round, which also bounds the context-owned tables (final exponentiation: 4 KB, Gt::pow: 6.9 KB per
// pairing OF ONE SUB-LAUNCH, not of the batch).  BN254_OPT_ROUND_PAIRS overrides (experiments).
size_t bn_round_pairs(const bn254_ctx *c) { return (size_t)bn_opt(c, BN254_OPT_ROUND_PAIRS); }
// as few sub-launches as possible with none above one round, all of (nearly) the same size: a ragged tail of a few pairings
// would cost a whole kernel latency (one wave takes as long as a full machine)
size_t bn_sub_launch(const bn254_ctx *c, size_t n) {
    const size_t round = bn_round_pairs(c);
    const size_t parts = (n + round - 1) / round;
    return parts <= 1 ? n : ((n + parts - 1) / parts + 31) / 32 * 32;
}

// Up to this many pairings (or Miller loops whose value only meets a final exponentiation) per call run ONE PER WAVE - the whole
// pairing as a program of the wave machine - instead of one per lane pair (4.2 ms whatever the count): BN254_OPT_WAVE_PAIRING_MAX.
static size_t bn_wave_pairing_max(const bn254_ctx *c) { return (size_t)bn_opt(c, BN254_OPT_WAVE_PAIRING_MAX); }
// naf: the value is only consumed by a final exponentiation, so the shorter NAF schedule may be used (pairing.hpp)
int bn_launch_miller(bn254_ctx *c, const void *p, const void *q, void *f, size_t n, hipStream_t s, bool naf) {
    if (naf && n <= bn_wave_pairing_max(c)) {
        BnScope sc(c, s, "miller_wave");
        return bn254_launch_pairing_W(p, q, f, n, 0, s);
    }
    // between the one-per-wave and the lane-pair regime: four lanes per pairing (bn254_kernels_q.hip) - while lane pairs would leave
    // SIMDs empty (up to BN254_OPT_QUAD_MAX = 64 per CU: one wave per SIMD of quads) the split Fq12 arithmetic is 1.4 x faster
    if (naf && n <= (size_t)bn_opt(c, BN254_OPT_QUAD_MAX)) {
        BnScope sc(c, s, "miller_quad");
        return bn254_launch_miller_Q(p, q, f, n, s);
    }
    const size_t step = bn_sub_launch(c, n);
    for (size_t lo = 0; lo < n; lo += step) {
        const size_t cnt = n - lo < step ? n - lo : step;
        BnScope sc(c, s, "miller");
        int rc = bn254_launch_miller_B((const char *)p + lo * sizeof(bn_g1), (const char *)q + lo * sizeof(bn_g2), (char *)f + lo * sizeof(bn_gt), cnt, naf ? 1 : 0, s);
        if (rc) return rc;
    }
    return BN254_OK;
}
// Up to this many final exponentiations per call run ONE PER WAVE (bn254_kernels_w.hip: 0.48 ms up to 1024 - one wave per SIMD -,
// 0.69 ms at 2048, 1.4 ms at 4096, while a lane pair needs 1.97 ms for its serial chain whatever the count): BN254_OPT_WAVE_FE_MAX.
static size_t bn_wave_fe_max(const bn254_ctx *c) { return (size_t)bn_opt(c, BN254_OPT_WAVE_FE_MAX); }
// table: the caller's own table buffer (pipelined path: one per chunk in flight) or NULL for the context's (under a BnScratchGuard)
int bn_launch_final_exp(bn254_ctx *c, const void *f, void *out, size_t n, hipStream_t s, BnBuf *table) {
    if (n <= bn_wave_fe_max(c)) {
        BnScope sc(c, s, "final_exp_wave");
        return bn254_launch_final_exp_W(f, out, n, s);
    }
    if (n <= (size_t)bn_opt(c, BN254_OPT_QUAD_MAX)) {
        BnBuf *t = table ? table : &c->exp_tbl;
        int rc = t->reserve(bn254_final_exp_table_bytes_Q(n)); if (rc) return rc;
        BnScope sc(c, s, "final_exp_quad");
        return bn254_launch_final_exp_Q(f, out, n, t->p, s);
    }
    BnBuf *t = table ? table : &c->exp_tbl;
    const size_t step = bn_sub_launch(c, n);
    int rc = t->reserve(bn254_final_exp_table_bytes_B(step)); if (rc) return rc;       // ONE table, reused by every sub-launch (stream order)
    for (size_t lo = 0; lo < n; lo += step) {
        const size_t cnt = n - lo < step ? n - lo : step;
        BnScope sc(c, s, "final_exp");
        rc = bn254_launch_final_exp_B((const char *)f + lo * sizeof(bn_gt), (char *)out + lo * sizeof(bn_gt), cnt, t->p, s);
        if (rc) return rc;
    }
    return BN254_OK;
}
// reduces n Fq12 values at `in` to one at `out`; `tmp` >= bn_product_tmp_bytes(n).  ONE launch (lane chunks -> wave-cooperative fold ->
// arrival tree over the waves, bn254_kernels_w.hip).
// Shape of the one-launch product tree, read off profiles/r03p_product_shape_sweep.txt.  Three ways to multiply, three prices:
// a lane pair multiplies two values in ~20-30 us but 32 of them do so at once per wave (`chunk` values per lane pair, then `bfly`
// butterfly levels across the pairs of a wave); the wave machine multiplies two values in ~2.7 us but one product at a time
// (what is left of a wave's `per_wave` partial products); a level of the arrival tree across waves costs ~5 us (product + publish).
// Few values: many small waves (the tree's log2 beats the serial fold).  Many: full waves, about one wave per SIMD of groups.
struct ProductShape { unsigned chunk, per_wave, bfly; };
static ProductShape product_shape(const bn254_ctx *c, size_t n) {
    ProductShape ps;
    if (n <= 64) ps = {1u, 2u, 0u};
    else if (n <= 2048) ps = {1u, 4u, 0u};
    else if (n <= 8192) ps = {1u, 8u, 0u};
    else if (n <= 16384) ps = {1u, 32u, 2u};
    else if (n < 65536) ps = {2u, 32u, 2u};
    else ps = {(unsigned)((n + 32767) / 32768), 32u, 2u};
    const long ch = bn_opt(c, BN254_OPT_PRODUCT_CHUNK), pw = bn_opt(c, BN254_OPT_PRODUCT_PER_WAVE), bf = bn_opt(c, BN254_OPT_PRODUCT_BFLY);
    if (ch >= 1) ps.chunk = (unsigned)ch;
    if (pw >= 1) ps.per_wave = (unsigned)pw;
    if (bf >= 0) ps.bfly = (unsigned)bf;
    return ps;
}
int bn_launch_product(bn254_ctx *c, const void *in, size_t n, void *out, void *tmp, hipStream_t s) {
    size_t grid, sb, cw;
    const ProductShape ps = product_shape(c, n);
    bn254_gt_reduce_sizes_W(n, ps.chunk, ps.per_wave, &grid, &sb, &cw);
    BnScope sc(c, s, "gt_product");
    return bn254_launch_gt_reduce_W(in, n, ps.chunk, ps.per_wave, ps.bfly, tmp, (char *)tmp + sb, out, s);
}
// out = final_exponentiation(in[0] * ... * in[m-1]): the tail of a sharded multi-pairing (the partial products of the ranks, then
// the ONE final exponentiation).  Up to 16 values: one wave-cooperative launch (15 products in a row cost what the second launch
// and the tree's levels would); more: product tree, then the exponentiation.
int bn_launch_product_final_exp(bn254_ctx *c, const void *in, size_t m, void *out, hipStream_t s) {
    if (m >= 1 && m <= 16) {
        BnScope sc(c, s, "gt_tail");
        return bn254_launch_gt_tail_W(in, 1, (unsigned)m, out, 1, s);
    }
    int rc = c->ws.reserve(bn_product_tmp_bytes(c, m)); if (rc) return rc;
    if ((rc = bn_launch_product(c, in, m, out, c->ws.p, s))) return rc;
    return bn_launch_final_exp(c, out, out, 1, s, nullptr);
}
size_t bn_product_tmp_bytes(const bn254_ctx *c, size_t n) {
    size_t grid, sb, cw;
    const ProductShape ps = product_shape(c, n ? n : 1);
    bn254_gt_reduce_sizes_W(n ? n : 1, ps.chunk, ps.per_wave, &grid, &sb, &cw);
    const size_t a = 2 * ((n + 3) / 4) * 384 + 384, b = sb + cw * sizeof(uint32_t) + 256;
    return a > b ? a : b;
}

// sub-launches of at most `step` units: fn(lo, cnt) enqueues one
template <class Fn>
static int bn_for_parts(size_t n, size_t step, Fn fn) {
    for (size_t lo = 0; lo < n; lo += step) {
        int rc = fn(lo, n - lo < step ? n - lo : step);
        if (rc) return rc;
    }
    return BN254_OK;
}

// out[i] = pairing(p[i], q[i]).  Small batches: Miller loop + final exponentiation per WAVE, one launch; otherwise the lane-pair
// kernels, the Miller values written to `out` and exponentiated in place (same 384-byte slots).
int bn_launch_pairing(bn254_ctx *c, const void *p, const void *q, void *out, size_t n, hipStream_t s, BnBuf *table) {
    if (n <= bn_wave_pairing_max(c)) {
        BnScope sc(c, s, "pairing_wave");
        return bn254_launch_pairing_W(p, q, out, n, 1, s);
    }
    int rc = bn_launch_miller(c, p, q, out, n, s, true); if (rc) return rc;
    return bn_launch_final_exp(c, out, out, n, s, table);
}

// The normalising kernels keep every lane's window table (640 B) in a buffer the sub-launches reuse.  Sub-launches of 2^20 lanes (671 MB; rounds
// 2-5: 2^18): these kernels run three resident waves per SIMD under plain oldest-first arbitration, a launch ends with every SIMD draining its last
// wave alone, and that tail is paid once per launch - 2^20 G1 multiplications in ONE launch of 16 waves per SIMD: 91.1 against 86.7 M/s in four
// launches on the same box, G2 +2 % (profiles/r06_ab_mul_launch_size.txt).
constexpr size_t BN_MUL_LANES_PER_LAUNCH = (size_t)1 << 20;
int bn_mul_dev(bn254_ctx *ctx, int g, const void *d_p, const void *d_k, void *d_out, size_t n, hipStream_t s, int normalize, BnBuf *table) {
    const size_t ps = g == 1 ? sizeof(bn_g1) : sizeof(bn_g2);
    const size_t step = normalize ? BN_MUL_LANES_PER_LAUNCH / (g == 1 ? 1 : 2) : BN_LAUNCH_MAX;
    BnBuf *t = table ? table : &ctx->mul_tbl;
    if (normalize) { int rc = t->reserve(bn254_mul_table_bytes_M(g, n < step ? n : step)); if (rc) return rc; }
    return bn_for_parts(n, step, [&](size_t lo, size_t cnt) -> int {
        const void *p = (const char *)d_p + lo * ps, *k = (const char *)d_k + lo * sizeof(bn_fr);
        void *o = (char *)d_out + lo * ps;
        BnScope sc(ctx, s, g == 1 ? "g1_mul" : "g2_mul");
        // registers-resident chains; G2 in the lane-pair mapping (bn254_kernels_mul.hip)
        return g == 1 ? bn254_launch_g1_mul_M(p, k, o, cnt, normalize, t->p, s) : bn254_launch_g2_mul_M(p, k, o, cnt, normalize, t->p, s);
    });
}

extern "C" {

int bn254_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int bn254_ctx_create(int device, bn254_ctx **out) {
    if (!out) return BN254_E_BAD_ARG;
    int n = bn254_device_count();
    if (n <= 0 || device < 0 || device >= n) return BN254_E_NO_DEVICE;
    BnDeviceGuard dev_guard;
    HIP_TRY(hipSetDevice(device));
    bn254_ctx *c = new (std::nothrow) bn254_ctx();
    if (!c) return BN254_E_ALLOC;
    c->device = device;
    if (hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) c->cus = 256;
    for (int k = 0; k < BN254_OPT_COUNT_; ++k) c->opt[k].store(k >= 1 && bn_opt_valid(k, bn_debug_env().opt[k]) ? bn_debug_env().opt[k] : -1);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return (int)e; }
    *out = c;
    return BN254_OK;
}
void bn254_ctx_destroy(bn254_ctx *c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> lk(g_default_mu);          // a default context handed out by bn_get_ctx must not dangle
        for (auto &d : g_default) if (d == c) d = nullptr;
    }
    BnDeviceGuard dev_guard;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (auto &r : c->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    c->ws.release(); c->exp_tbl.release(); c->pow_tbl.release(); c->miller_state.release(); c->mul_tbl.release();
    for (auto &b : c->stage) b.release();
    for (auto &s : c->slot) {
        for (auto &b : s.d_in) b.release();
        s.d_out.release(); s.tbl.release();
        if (s.stream) hipStreamDestroy(s.stream);
    }
    if (c->scratch_ev) hipEventDestroy(c->scratch_ev);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}
const char *bn254_error_string(int code) {
    switch (code) {
        case BN254_OK: return "ok";
        case BN254_E_NO_DEVICE: return "no usable HIP device (this engine has no CPU fallback)";
        case BN254_E_BAD_ARG: return "bad argument";
        case BN254_E_ALLOC: return "device allocation failed";
        case BN254_E_COMM: return "RCCL / peer exchange failed";
        case BN254_E_INTERNAL: return "internal error (C++ exception stopped at the C boundary)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}
// Kept for ABI compatibility with rounds 1-4: the one-lane-per-pairing mapping (0) was a test double and left the library in round 5
// (tests/testdouble/); only 1 - the lane-pair mapping with its wave and four-lane siblings - is accepted.
int bn254_ctx_set_mapping(bn254_ctx *ctx, int mapping) {
    if (!ctx || mapping != 1) return BN254_E_BAD_ARG;
    return BN254_OK;
}
int bn254_ctx_set_option(bn254_ctx *ctx, int key, long value) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (key < 1 || key >= BN254_OPT_COUNT_ || (value >= 0 && !bn_opt_valid(key, value))) return BN254_E_BAD_ARG;
    ctx->opt[key].store(value < 0 ? -1 : value);
    return BN254_OK;
}
int bn254_ctx_get_option(bn254_ctx *ctx, int key, long *value) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (key < 1 || key >= BN254_OPT_COUNT_ || !value) return BN254_E_BAD_ARG;
    *value = bn_opt(ctx, key);
    return BN254_OK;
}

int bn254_ctx_get_option_raw(bn254_ctx *ctx, int key, long *value) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (key < 1 || key >= BN254_OPT_COUNT_ || !value) return BN254_E_BAD_ARG;
    const long v = ctx->opt[key].load(std::memory_order_relaxed);
    *value = v < 0 ? -1 : v;
    return BN254_OK;
}

// ---------------------------------------------------------------------------------------------- device-resident API
constexpr size_t BN_N_MAX = (size_t)1 << 40;          // sanity bound on a batch; launches are cut to size internally
#define BN_DEV_PROLOGUE(null_check, limit)                                           \
    int rc = bn_get_ctx(ctx); if (rc) return rc;                                     \
    if (n == 0) return BN254_OK;                                                     \
    if ((null_check) || n > (limit)) return BN254_E_BAD_ARG;                         \
    BnDeviceGuard dev_guard;                                                         \
    HIP_TRY(hipSetDevice(ctx->device));                                              \
    hipStream_t s = (hipStream_t)stream

int bn254_miller_batch_dev(bn254_ctx *ctx, const void *d_p, const void *d_q, void *d_f, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_p || !d_q || !d_f, BN_N_MAX);
    return bn_launch_miller(ctx, d_p, d_q, d_f, n, s, false);
}
int bn254_final_exp_batch_dev(bn254_ctx *ctx, const void *d_f, void *d_out, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_f || !d_out, BN_N_MAX);
    BnScratchGuard g(ctx, s); if (g.rc) return g.rc;
    return bn_launch_final_exp(ctx, d_f, d_out, n, s, nullptr);
}
int bn254_pairing_batch_dev(bn254_ctx *ctx, const void *d_p, const void *d_q, void *d_out, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_p || !d_q || !d_out, BN_N_MAX);
    BnScratchGuard g(ctx, s); if (g.rc) return g.rc;
    return bn_launch_pairing(ctx, d_p, d_q, d_out, n, s, nullptr);
}
int bn254_gt_product_dev(bn254_ctx *ctx, const void *d_in, size_t n, void *d_out, void *stream) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (!d_out || (n && !d_in) || n > 0x7fffffffu / 96) return BN254_E_BAD_ARG;      // one launch: 32-bit word offsets in the kernel
    BnDeviceGuard dev_guard;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {       // empty product = one
        bn_gt one; memset(&one, 0, sizeof one);
        one.c[0] = 0xd35d438dc58f0d9dull; one.c[1] = 0x0a78eb28f5c70b3dull; one.c[2] = 0x666ea36f7879462cull; one.c[3] = 0x0e0a77c19a07df2full;
        HIP_TRY(hipMemcpyAsync(d_out, &one, sizeof one, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        return BN254_OK;
    }
    BnScratchGuard g(ctx, s); if (g.rc) return g.rc;
    rc = ctx->ws.reserve(bn_product_tmp_bytes(ctx, n)); if (rc) return rc;
    return bn_launch_product(ctx, d_in, n, d_out, ctx->ws.p, s);
}
int bn254_gt_product_final_exp_dev(bn254_ctx *ctx, const void *d_in, size_t m, void *d_out, void *stream) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (!d_out || !d_in || m == 0 || m > BN_N_MAX) return BN254_E_BAD_ARG;
    BnDeviceGuard dev_guard;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    BnScratchGuard g(ctx, s); if (g.rc) return g.rc;
    return bn_launch_product_final_exp(ctx, d_in, m, d_out, s);
}
// How many pairs share one accumulator f in the multi-pairing's Miller loop (pairing.hpp miller_loop_shared): as many as keep at least
// one full machine round of lane pairs busy - 4 from four rounds of pairs on (configs[3] on one GPU: 2^18), 2 from two, else the
// plain kernel (a per-GPU shard of 2^15 must not be folded onto a quarter of the machine).  BN254_OPT_MILLER_SHARED = 1|2|4 overrides.
static int miller_shared_m(const bn254_ctx *c, size_t n) {
    const long forced = bn_opt(c, BN254_OPT_MILLER_SHARED);
    if (forced == 1 || forced == 2 || forced == 4) return (int)forced;
    const size_t round = bn_round_pairs(c);
    return n >= 4 * round ? 4 : n >= 2 * round ? 2 : 1;
}
int bn254_miller_product_dev(bn254_ctx *ctx, const void *d_p, const void *d_q, size_t n, void *d_partial, void *stream) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (!d_partial || (n && (!d_p || !d_q)) || n > 0x7fffffffu / 96) return BN254_E_BAD_ARG;      // (the Miller values of all n meet ONE product launch)
    if (n == 0) return bn254_gt_product_dev(ctx, nullptr, 0, d_partial, stream);
    BnDeviceGuard dev_guard;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    BnScratchGuard g(ctx, s); if (g.rc) return g.rc;
    const int m = (n > bn_wave_pairing_max(ctx)) ? miller_shared_m(ctx, n) : 1;
    const size_t nv = (n + (size_t)m - 1) / (size_t)m;             // Miller values that reach the product tree
    const size_t fbytes = nv * 384;
    rc = ctx->ws.reserve(fbytes + bn_product_tmp_bytes(ctx, nv)); if (rc) return rc;
    if (m == 1) {
        rc = bn_launch_miller(ctx, d_p, d_q, ctx->ws.p, n, s, true); if (rc) return rc;
    } else {
        // sub-launches of at most one round of LANE PAIRS, each with m pairs: one state buffer, reused in stream order
        const size_t step = bn_sub_launch(ctx, nv);
        rc = ctx->miller_state.reserve(bn254_miller_shared_state_bytes_B(step * (size_t)m, m)); if (rc) return rc;
        for (size_t lo = 0; lo < nv; lo += step) {
            const size_t groups = nv - lo < step ? nv - lo : step, first = lo * (size_t)m;
            const size_t cnt = n - first < groups * (size_t)m ? n - first : groups * (size_t)m;
            BnScope sc(ctx, s, "miller_shared");
            rc = bn254_launch_miller_shared_B((const char *)d_p + first * sizeof(bn_g1), (const char *)d_q + first * sizeof(bn_g2),
                                              (char *)ctx->ws.p + lo * sizeof(bn_gt), cnt, m, ctx->miller_state.p, s);
            if (rc) return rc;
        }
    }
    return bn_launch_product(ctx, ctx->ws.p, nv, d_partial, (char *)ctx->ws.p + fbytes, s);
}
static int mul_dev(bn254_ctx *ctx, int g, const void *d_p, const void *d_k, void *d_out, size_t n, void *stream, int normalize) {
    BN_DEV_PROLOGUE(!d_p || !d_k || !d_out, BN_N_MAX);
    BnScratchGuard gd(ctx, s); if (gd.rc) return gd.rc;          // the window tables are context-owned scratch
    return bn_mul_dev(ctx, g, d_p, d_k, d_out, n, s, normalize);
}
int bn254_g1_mul_batch_dev(bn254_ctx *c, const void *p, const void *k, void *o, size_t n, void *s) { return mul_dev(c, 1, p, k, o, n, s, 1); }
int bn254_g2_mul_batch_dev(bn254_ctx *c, const void *p, const void *k, void *o, size_t n, void *s) { return mul_dev(c, 2, p, k, o, n, s, 1); }
int bn254_g1_mul_jacobian_dev(bn254_ctx *c, const void *p, const void *k, void *o, size_t n, void *s) { return mul_dev(c, 1, p, k, o, n, s, 0); }
int bn254_g2_mul_jacobian_dev(bn254_ctx *c, const void *p, const void *k, void *o, size_t n, void *s) { return mul_dev(c, 2, p, k, o, n, s, 0); }

int bn254_g2_precompute_dev(bn254_ctx *ctx, const void *d_q, void *d_coeffs, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_q || !d_coeffs, 0x7fffffffu / (102 * 48));
    BnScope sc(ctx, s, "g2_precompute");
    return bn254_launch_g2_precompute_B(d_q, d_coeffs, n, s);
}
int bn254_miller_prepared_dev(bn254_ctx *ctx, const void *d_p, const void *d_coeffs, int shared, void *d_f, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_p || !d_coeffs || !d_f, 0x7fffffffu / (102 * 48));
    BnScope sc(ctx, s, "miller_prepared");
    return bn254_launch_miller_prepared_B(d_p, d_coeffs, shared, d_f, n, s);
}
// ---- native prepared-G2 mode (include/bn254_hip.h): the handle owns its table; one launch addresses it with 32-bit columns (2 per point)
constexpr size_t BN_PREPARED_MAX = (size_t)1 << 22;
int bn254_g2_prepare_dev(bn254_ctx *ctx, const void *d_q, size_t nq, bn254_g2_prepared **out, void *stream) {
    if (!out) return BN254_E_BAD_ARG;
    *out = nullptr;
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (!d_q || nq == 0 || nq > BN_PREPARED_MAX) return BN254_E_BAD_ARG;
    BnDeviceGuard dev_guard;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    bn254_g2_prepared *h = new (std::nothrow) bn254_g2_prepared();
    if (!h) return BN254_E_ALLOC;
    h->device = ctx->device; h->nq = nq; h->bytes = bn254_native_table_bytes_B(nq);
    if (hipMalloc(&h->table, h->bytes) != hipSuccess) { delete h; return BN254_E_ALLOC; }
    if (hipMalloc(&h->inf, nq * sizeof(uint32_t)) != hipSuccess) { hipFree(h->table); delete h; return BN254_E_ALLOC; }
    {
        BnScope sc(ctx, s, "g2_prepare_native");
        rc = bn254_launch_g2_prepare_native_B(d_q, h->table, h->inf, nq, s);
    }
    // the points themselves stay with the handle: a SMALL call is served by the general path's one-pairing-per-wave kernels (1.0 ms for up to
    // 1024 pairings, 1.95 ms at 3072 - the native Miller loop is a lane-pair kernel and needs 1.7 ms however few pairings there are:
    // tools/prepared_latency.py).  One shared point is repeated small_max times by doubling copies.
    h->small_max = (size_t)12 * (size_t)(ctx->cus > 0 ? ctx->cus : 256);
    const size_t qn = nq == 1 ? h->small_max : nq;
    if (!rc && hipMalloc(&h->q, qn * sizeof(bn_g2)) != hipSuccess) rc = BN254_E_ALLOC;
    if (!rc && hipMemcpyAsync(h->q, d_q, nq * sizeof(bn_g2), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = BN254_E_INTERNAL;
    if (!rc && nq == 1)
        for (size_t have = 1; have < qn && !rc; have *= 2) {
            const size_t cnt = have < qn - have ? have : qn - have;
            if (hipMemcpyAsync((char *)h->q + have * sizeof(bn_g2), h->q, cnt * sizeof(bn_g2), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = BN254_E_INTERNAL;
        }
    if (rc) { hipFree(h->table); hipFree(h->inf); if (h->q) hipFree(h->q); delete h; return rc; }
    *out = h;
    return BN254_OK;
}
void bn254_g2_prepared_destroy(bn254_g2_prepared *h) {
    if (!h) return;
    BnDeviceGuard dev_guard;
    hipSetDevice(h->device);
    hipFree(h->table); hipFree(h->inf); hipFree(h->q);
    delete h;
}
size_t bn254_g2_prepared_count(const bn254_g2_prepared *h) { return h ? h->nq : 0; }
size_t bn254_g2_prepared_bytes(const bn254_g2_prepared *h) { return h ? h->bytes + h->nq * sizeof(uint32_t) + (h->nq == 1 ? h->small_max : h->nq) * sizeof(bn_g2) : 0; }
// p[i] against point (nq == 1 ? 0 : q_first + i): sub-launches of at most one machine round, like the fused Miller loop
static int bn_launch_miller_native(bn254_ctx *c, const void *p, const bn254_g2_prepared *h, size_t q_first, void *f, size_t n, hipStream_t s) {
    const int shared = h->nq == 1;
    const size_t step = bn_sub_launch(c, n);
    for (size_t lo = 0; lo < n; lo += step) {
        const size_t cnt = n - lo < step ? n - lo : step;
        BnScope sc(c, s, "miller_native");
        int rc = bn254_launch_miller_native_B((const char *)p + lo * sizeof(bn_g1), h->table, h->inf, h->nq, shared ? 0 : q_first + lo, shared, (char *)f + lo * sizeof(bn_gt), cnt, s);
        if (rc) return rc;
    }
    return BN254_OK;
}
#define BN_PREP_CHECK()                                                                                              \
    if (!prep || prep->device != ctx->device || (prep->nq != 1 && (q_first > prep->nq || n > prep->nq - q_first))) return BN254_E_BAD_ARG
int bn254_miller_prepared_native_dev(bn254_ctx *ctx, const void *d_p, const bn254_g2_prepared *prep, size_t q_first, void *d_f, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_p || !d_f, BN_N_MAX);
    BN_PREP_CHECK();
    return bn_launch_miller_native(ctx, d_p, prep, q_first, d_f, n, s);
}
int bn254_pairing_prepared_native_batch_dev(bn254_ctx *ctx, const void *d_p, const bn254_g2_prepared *prep, size_t q_first, void *d_out, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_p || !d_out, BN_N_MAX);
    BN_PREP_CHECK();
    BnScratchGuard g(ctx, s); if (g.rc) return g.rc;
    // small calls: the whole pairing per WAVE on the points kept with the handle (same bytes out; BN254_OPT_WAVE_PAIRING_MAX = 0 turns this off)
    const size_t small = prep->small_max < bn_wave_pairing_max(ctx) ? prep->small_max : bn_wave_pairing_max(ctx);
    if (n <= small) return bn_launch_pairing(ctx, d_p, (const char *)prep->q + (prep->nq == 1 ? 0 : q_first) * sizeof(bn_g2), d_out, n, s, nullptr);
    rc = bn_launch_miller_native(ctx, d_p, prep, q_first, d_out, n, s); if (rc) return rc;
    return bn_launch_final_exp(ctx, d_out, d_out, n, s, nullptr);
}
// local part of a multi-pairing over prepared points: prod_i miller(p[i], point q_first + i), un-exponentiated.  The shared-accumulator kernels
// (pairing.hpp miller_loop_native_shared) from two machine rounds of pairs on, like the fused path (miller_shared_m); a small call takes the
// general path on the points kept with the handle.
int bn254_miller_product_prepared_native_dev(bn254_ctx *ctx, const void *d_p, const bn254_g2_prepared *prep, size_t q_first, size_t n, void *d_partial, void *stream) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (!d_partial || (n && !d_p) || n > 0x7fffffffu / 96) return BN254_E_BAD_ARG;
    if (n == 0) return bn254_gt_product_dev(ctx, nullptr, 0, d_partial, stream);
    BN_PREP_CHECK();
    const size_t small = prep->small_max < bn_wave_pairing_max(ctx) ? prep->small_max : bn_wave_pairing_max(ctx);
    if (n <= small) return bn254_miller_product_dev(ctx, d_p, (const char *)prep->q + (prep->nq == 1 ? 0 : q_first) * sizeof(bn_g2), n, d_partial, stream);
    BnDeviceGuard dev_guard;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    BnScratchGuard g(ctx, s); if (g.rc) return g.rc;
    const int m = miller_shared_m(ctx, n);
    const size_t nv = (n + (size_t)m - 1) / (size_t)m;             // Miller values that reach the product tree
    const size_t fbytes = nv * 384;
    rc = ctx->ws.reserve(fbytes + bn_product_tmp_bytes(ctx, nv)); if (rc) return rc;
    if (m == 1) {
        rc = bn_launch_miller_native(ctx, d_p, prep, q_first, ctx->ws.p, n, s); if (rc) return rc;
    } else {
        const int shared = prep->nq == 1;
        const size_t step = bn_sub_launch(ctx, nv);                // sub-launches of at most one round of LANE PAIRS, each with m pairs
        for (size_t lo = 0; lo < nv; lo += step) {
            const size_t groups = nv - lo < step ? nv - lo : step, first = lo * (size_t)m;
            const size_t cnt = n - first < groups * (size_t)m ? n - first : groups * (size_t)m;
            BnScope sc(ctx, s, "miller_native_shared");
            rc = bn254_launch_miller_native_shared_B((const char *)d_p + first * sizeof(bn_g1), prep->table, prep->inf, prep->nq, shared ? 0 : q_first + first, shared,
                                                     (char *)ctx->ws.p + lo * sizeof(bn_gt), cnt, m, s);
            if (rc) return rc;
        }
    }
    return bn_launch_product(ctx, ctx->ws.p, nv, d_partial, (char *)ctx->ws.p + fbytes, s);
}
int bn254_gt_mul_batch_dev(bn254_ctx *ctx, const void *d_a, const void *d_b, void *d_out, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_a || !d_b || !d_out, BN_N_MAX);
    return bn_for_parts(n, BN_LAUNCH_MAX, [&](size_t lo, size_t cnt) -> int {
        BnScope sc(ctx, s, "gt_mul");
        return bn254_launch_gt_mul_B((const char *)d_a + lo * sizeof(bn_gt), (const char *)d_b + lo * sizeof(bn_gt), (char *)d_out + lo * sizeof(bn_gt), cnt, s);
    });
}
int bn254_gt_pow_batch_dev(bn254_ctx *ctx, const void *d_a, const void *d_k, void *d_out, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_a || !d_k || !d_out, BN_N_MAX);
    BnScratchGuard g(ctx, s); if (g.rc) return g.rc;
    // ONE window table (33 x 224 B per lane of a sub-launch), reused by every sub-launch in stream order: 970 MB at a full round
    // whatever the batch size (round 2 allocated 6.9 KB x n)
    const size_t step = bn_sub_launch(ctx, n);
    rc = ctx->pow_tbl.reserve(bn254_gt_pow_table_bytes_B(step)); if (rc) return rc;
    return bn_for_parts(n, step, [&](size_t lo, size_t cnt) -> int {
        BnScope sc(ctx, s, "gt_pow");
        return bn254_launch_gt_pow_B((const char *)d_a + lo * sizeof(bn_gt), (const char *)d_k + lo * sizeof(bn_fr), (char *)d_out + lo * sizeof(bn_gt), cnt, ctx->pow_tbl.p, (int)bn_opt(ctx, BN254_OPT_GT_POW_MODE), s);
    });
}
int bn254_gt_inverse_batch_dev(bn254_ctx *ctx, const void *d_a, void *d_out, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_a || !d_out, BN_N_MAX);
    return bn_for_parts(n, BN_LAUNCH_MAX, [&](size_t lo, size_t cnt) -> int {
        BnScope sc(ctx, s, "gt_inverse");
        return bn254_launch_gt_inverse_B((const char *)d_a + lo * sizeof(bn_gt), (char *)d_out + lo * sizeof(bn_gt), cnt, s);
    });
}
int bn254_exp_by_neg_z_dev(bn254_ctx *ctx, const void *d_a, void *d_out, size_t n, void *stream) {
    BN_DEV_PROLOGUE(!d_a || !d_out, BN_N_MAX);
    return bn_for_parts(n, BN_LAUNCH_MAX, [&](size_t lo, size_t cnt) -> int {
        BnScope sc(ctx, s, "exp_by_neg_z");
        return bn254_launch_exp_by_neg_z_B((const char *)d_a + lo * sizeof(bn_gt), (char *)d_out + lo * sizeof(bn_gt), cnt, s);
    });
}

// ---------------------------------------------------------------------------------------------- host-buffer API
// Every function below holds the context's mutex for the whole call: concurrent callers of one context (in particular of
// the default context behind ctx == NULL) are serialised, never interleaved on the staging memory.
#define BN_HOST_PROLOGUE()                                                           \
    int rc = bn_get_ctx(ctx); if (rc) return rc;                                     \
    std::lock_guard<std::mutex> host_lock(ctx->mu);                                  \
    BnDeviceGuard dev_guard;                                                         \
    HIP_TRY(hipSetDevice(ctx->device))

int bn254_pairing_batch(bn254_ctx *ctx, const bn_g1 *p, const bn_g2 *q, bn_gt *out, size_t n) {
    if (n == 0) return BN254_OK;
    if (!p || !q || !out || n > BN_N_MAX) return BN254_E_BAD_ARG;
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    BnDeviceGuard dev_guard;                 // concurrency is arbitrated per pipeline slot (BnSlotLease), not by the context mutex
    HIP_TRY(hipSetDevice(ctx->device));
    return bn_no_throw([&] { return bn_pairing_batch_pipelined(ctx, p, q, out, n); });
}
int bn254_pairing_product(bn254_ctx *ctx, const bn_g1 *p, const bn_g2 *q, size_t n, bn_gt *out) {
    if (!out || (n && (!p || !q)) || n > 0x7fffffffu / 96) return BN254_E_BAD_ARG;
    BN_HOST_PROLOGUE();
    BnBuf &dp = ctx->stage[0], &dq = ctx->stage[1], &dpart = ctx->stage[2];
    if ((rc = dp.reserve(n * sizeof(bn_g1))) || (rc = dq.reserve(n * sizeof(bn_g2))) || (rc = dpart.reserve(sizeof(bn_gt)))) return rc;
    if (n) {
        HIP_TRY(hipMemcpyAsync(dp.p, p, n * sizeof(bn_g1), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dq.p, q, n * sizeof(bn_g2), hipMemcpyHostToDevice, ctx->stream));
    }
    // FE is a homomorphism: FE(prod miller_i) = prod FE(miller_i); one final exponentiation for the whole product
    rc = bn254_miller_product_dev(ctx, dp.p, dq.p, n, dpart.p, ctx->stream); if (rc) return rc;
    rc = bn254_final_exp_batch_dev(ctx, dpart.p, dpart.p, 1, ctx->stream); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dpart.p, sizeof(bn_gt), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BN254_OK;
}
int bn254_g1_mul_batch(bn254_ctx *ctx, const bn_g1 *p, const bn_fr *k, bn_g1 *out, size_t n) {
    if (n == 0) return BN254_OK;
    if (!p || !k || !out || n > BN_N_MAX) return BN254_E_BAD_ARG;
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    BnDeviceGuard dev_guard;                 // concurrency is arbitrated per pipeline slot (BnSlotLease), not by the context mutex
    HIP_TRY(hipSetDevice(ctx->device));
    return bn_no_throw([&] { return bn_mul_batch_pipelined(ctx, 1, p, k, out, n); });
}
int bn254_g2_mul_batch(bn254_ctx *ctx, const bn_g2 *p, const bn_fr *k, bn_g2 *out, size_t n) {
    if (n == 0) return BN254_OK;
    if (!p || !k || !out || n > BN_N_MAX) return BN254_E_BAD_ARG;
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    BnDeviceGuard dev_guard;                 // concurrency is arbitrated per pipeline slot (BnSlotLease), not by the context mutex
    HIP_TRY(hipSetDevice(ctx->device));
    return bn_no_throw([&] { return bn_mul_batch_pipelined(ctx, 2, p, k, out, n); });
}
int bn254_g2_precompute(bn254_ctx *ctx, const bn_g2 *q, bn_ell_coeffs *coeffs, size_t n) {
    if (n == 0) return BN254_OK;
    if (!q || !coeffs) return BN254_E_BAD_ARG;
    BN_HOST_PROLOGUE();
    BnBuf &dq = ctx->stage[0], &dc = ctx->stage[1];
    size_t cb = n * 102 * sizeof(bn_ell_coeffs);
    if ((rc = dq.reserve(n * sizeof(bn_g2))) || (rc = dc.reserve(cb))) return rc;
    HIP_TRY(hipMemcpyAsync(dq.p, q, n * sizeof(bn_g2), hipMemcpyHostToDevice, ctx->stream));
    rc = bn254_g2_precompute_dev(ctx, dq.p, dc.p, n, ctx->stream); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(coeffs, dc.p, cb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BN254_OK;
}
int bn254_pairing_prepared_batch(bn254_ctx *ctx, const bn_g1 *p, const bn_ell_coeffs *coeffs, int shared, bn_gt *out, size_t n) {
    if (n == 0) return BN254_OK;
    if (!p || !coeffs || !out) return BN254_E_BAD_ARG;
    BN_HOST_PROLOGUE();
    BnBuf &dp = ctx->stage[0], &dc = ctx->stage[1], &dout = ctx->stage[2];
    size_t cb = (shared ? 1 : n) * 102 * sizeof(bn_ell_coeffs);
    if ((rc = dp.reserve(n * sizeof(bn_g1))) || (rc = dc.reserve(cb)) || (rc = dout.reserve(n * sizeof(bn_gt)))) return rc;
    HIP_TRY(hipMemcpyAsync(dp.p, p, n * sizeof(bn_g1), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dc.p, coeffs, cb, hipMemcpyHostToDevice, ctx->stream));
    rc = bn254_miller_prepared_dev(ctx, dp.p, dc.p, shared, dout.p, n, ctx->stream); if (rc) return rc;
    rc = bn254_final_exp_batch_dev(ctx, dout.p, dout.p, n, ctx->stream); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout.p, n * sizeof(bn_gt), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BN254_OK;
}
int bn254_g2_prepare(bn254_ctx *ctx, const bn_g2 *q, size_t nq, bn254_g2_prepared **out) {
    if (!out) return BN254_E_BAD_ARG;
    *out = nullptr;
    if (!q || nq == 0 || nq > BN_PREPARED_MAX) return BN254_E_BAD_ARG;
    BN_HOST_PROLOGUE();
    BnBuf &dq = ctx->stage[0];
    if ((rc = dq.reserve(nq * sizeof(bn_g2)))) return rc;
    HIP_TRY(hipMemcpyAsync(dq.p, q, nq * sizeof(bn_g2), hipMemcpyHostToDevice, ctx->stream));
    rc = bn254_g2_prepare_dev(ctx, dq.p, nq, out, ctx->stream); if (rc) return rc;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { bn254_g2_prepared_destroy(*out); *out = nullptr; return (int)e; }
    return BN254_OK;
}
int bn254_g2_prepared_export(bn254_ctx *ctx, const bn254_g2_prepared *prep, void *host_table, size_t bytes) {
    if (!prep || !host_table || bytes != prep->nq * (size_t)BN254_PREPARED_NATIVE_BYTES) return BN254_E_BAD_ARG;
    BN_HOST_PROLOGUE();
    if (prep->device != ctx->device) return BN254_E_BAD_ARG;
    // the device table ends with the identity column pair (bn254_kernels_b.hip): every row of 2 nq + 2 columns gives up its first 2 nq
    const size_t row = 2 * prep->nq * 16, rows = bytes / row;
    HIP_TRY(hipMemcpy2DAsync(host_table, row, prep->table, row + 32, row, rows, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BN254_OK;
}
int bn254_pairing_prepared_native_batch(bn254_ctx *ctx, const bn_g1 *p, const bn254_g2_prepared *prep, bn_gt *out, size_t n) {
    if (n == 0) return BN254_OK;
    if (!p || !prep || !out || n > BN_N_MAX) return BN254_E_BAD_ARG;
    BN_HOST_PROLOGUE();
    BnBuf &dp = ctx->stage[0], &dout = ctx->stage[2];
    if ((rc = dp.reserve(n * sizeof(bn_g1))) || (rc = dout.reserve(n * sizeof(bn_gt)))) return rc;
    HIP_TRY(hipMemcpyAsync(dp.p, p, n * sizeof(bn_g1), hipMemcpyHostToDevice, ctx->stream));
    rc = bn254_pairing_prepared_native_batch_dev(ctx, dp.p, prep, 0, dout.p, n, ctx->stream); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout.p, n * sizeof(bn_gt), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BN254_OK;
}
int bn254_pairing_product_prepared_native(bn254_ctx *ctx, const bn_g1 *p, const bn254_g2_prepared *prep, size_t n, bn_gt *out) {
    if (!out || !prep || (n && !p) || n > 0x7fffffffu / 96) return BN254_E_BAD_ARG;
    BN_HOST_PROLOGUE();
    BnBuf &dp = ctx->stage[0], &dpart = ctx->stage[2];
    if ((rc = dp.reserve(n * sizeof(bn_g1))) || (rc = dpart.reserve(sizeof(bn_gt)))) return rc;
    if (n) HIP_TRY(hipMemcpyAsync(dp.p, p, n * sizeof(bn_g1), hipMemcpyHostToDevice, ctx->stream));
    rc = bn254_miller_product_prepared_native_dev(ctx, dp.p, prep, 0, n, dpart.p, ctx->stream); if (rc) return rc;
    rc = bn254_final_exp_batch_dev(ctx, dpart.p, dpart.p, 1, ctx->stream); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dpart.p, sizeof(bn_gt), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BN254_OK;
}
// wire format: host buffers in, host buffers out
static int wire_host(bn254_ctx *ctx, int g, int decode, const void *in, void *out, int32_t *status, size_t n) {
    if (n == 0) return BN254_OK;
    if (!in || !out || (decode && !status) || n > 0x7fffffffu / 129) return BN254_E_BAD_ARG;
    BN_HOST_PROLOGUE();
    size_t ps = g == 0 ? sizeof(bn_fr) : g == 1 ? sizeof(bn_g1) : sizeof(bn_g2), rs = g == 0 ? BN254_FR_WIRE_BYTES : g == 1 ? BN254_G1_WIRE_BYTES : BN254_G2_WIRE_BYTES;
    size_t in_b = n * (decode ? rs : ps), out_b = n * (decode ? ps : rs);
    BnBuf &din = ctx->stage[0], &dout = ctx->stage[1], &dst = ctx->stage[2];
    if ((rc = din.reserve(in_b)) || (rc = dout.reserve(out_b)) || (rc = dst.reserve(n * sizeof(int32_t)))) return rc;
    HIP_TRY(hipMemcpyAsync(din.p, in, in_b, hipMemcpyHostToDevice, ctx->stream));
    {
        BnScope sc(ctx, ctx->stream, decode ? "wire_decode" : "wire_encode");
        dim3 grid(grid_for(n)), block(BLOCK);
        if (g == 0 && !decode) hipLaunchKernelGGL(bn254_fr_encode_k, grid, block, 0, ctx->stream, (const uint32_t *)din.p, (uint8_t *)dout.p, (uint32_t)n);
        if (g == 0 && decode) hipLaunchKernelGGL(bn254_fr_decode_k, grid, block, 0, ctx->stream, (const uint8_t *)din.p, (uint32_t *)dout.p, (int32_t *)dst.p, (uint32_t)n);
        if (g == 1 && !decode) hipLaunchKernelGGL(bn254_g1_encode_k, grid, block, 0, ctx->stream, (const uint32_t *)din.p, (uint8_t *)dout.p, (uint32_t)n);
        if (g == 2 && !decode) hipLaunchKernelGGL(bn254_g2_encode_k, grid, block, 0, ctx->stream, (const uint32_t *)din.p, (uint8_t *)dout.p, (uint32_t)n);
        if (g == 1 && decode) hipLaunchKernelGGL(bn254_g1_decode_k, grid, block, 0, ctx->stream, (const uint8_t *)din.p, (uint32_t *)dout.p, (int32_t *)dst.p, (uint32_t)n);
        if (g == 2 && decode) hipLaunchKernelGGL(bn254_g2_decode_k, grid, block, 0, ctx->stream, (const uint8_t *)din.p, (uint32_t *)dout.p, (int32_t *)dst.p, (uint32_t)n);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, dout.p, out_b, hipMemcpyDeviceToHost, ctx->stream));
    if (decode) HIP_TRY(hipMemcpyAsync(status, dst.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BN254_OK;
}
int bn254_fr_encode_batch(bn254_ctx *ctx, const bn_fr *k, uint8_t *out, size_t n) { return wire_host(ctx, 0, 0, k, out, nullptr, n); }
int bn254_fr_decode_batch(bn254_ctx *ctx, const uint8_t *in, bn_fr *out, int32_t *status, size_t n) { return wire_host(ctx, 0, 1, in, out, status, n); }
int bn254_g1_encode_batch(bn254_ctx *ctx, const bn_g1 *p, uint8_t *out, size_t n) { return wire_host(ctx, 1, 0, p, out, nullptr, n); }
int bn254_g2_encode_batch(bn254_ctx *ctx, const bn_g2 *p, uint8_t *out, size_t n) { return wire_host(ctx, 2, 0, p, out, nullptr, n); }
int bn254_g1_decode_batch(bn254_ctx *ctx, const uint8_t *in, bn_g1 *out, int32_t *status, size_t n) { return wire_host(ctx, 1, 1, in, out, status, n); }
int bn254_g2_decode_batch(bn254_ctx *ctx, const uint8_t *in, bn_g2 *out, int32_t *status, size_t n) { return wire_host(ctx, 2, 1, in, out, status, n); }
// ---- the crate's actual byte STREAM (groups/mod.rs:143-205): a point at infinity is the lone byte 0, a finite point is 4 followed by
// its coordinates - records of variable length.  The stream is cut into records on the host (a tag decides the length), the fixed
// records go through the batch kernels above.
static int stream_encode(bn254_ctx *ctx, int g, const void *p, size_t n, uint8_t *out, size_t cap, size_t *written) {
    if (!written || (n && (!p || !out))) return BN254_E_BAD_ARG;
    const size_t rs = g == 1 ? BN254_G1_WIRE_BYTES : BN254_G2_WIRE_BYTES;
    return bn_no_throw([&]() -> int {
        std::vector<uint8_t> fixed(n * rs);
        int rc = g == 1 ? bn254_g1_encode_batch(ctx, (const bn_g1 *)p, fixed.data(), n) : bn254_g2_encode_batch(ctx, (const bn_g2 *)p, fixed.data(), n);
        if (rc) return rc;
        size_t w = 0;
        for (size_t i = 0; i < n; ++i) {
            const uint8_t *r = fixed.data() + i * rs;
            const size_t len = r[0] == 0 ? 1 : rs;
            if (w + len > cap) return BN254_E_BAD_ARG;
            memcpy(out + w, r, len);
            w += len;
        }
        *written = w;
        return BN254_OK;
    });
}
static int stream_decode(bn254_ctx *ctx, int g, const uint8_t *in, size_t len, void *out, int32_t *status, size_t max_points, size_t *count, size_t *consumed) {
    if (!count || !consumed || (len && !in) || (max_points && (!out || !status))) return BN254_E_BAD_ARG;
    const size_t rs = g == 1 ? BN254_G1_WIRE_BYTES : BN254_G2_WIRE_BYTES;
    { int rc0 = bn_get_ctx(ctx); if (rc0) return rc0; }
    return bn_no_throw([&]() -> int {
        std::vector<uint8_t> fixed;
        std::vector<size_t> ends;                                   // stream position behind every record
        size_t pos = 0, n = 0;
        while (pos < len && n < max_points) {
            // tag 0: one byte; tag 4: a full record; any other tag is the crate's "invalid leading byte" - it consumes the byte it read
            const size_t rec = in[pos] == 4 ? rs : 1;
            if (pos + rec > len) break;                         // truncated record: stop in front of it
            fixed.resize((n + 1) * rs, 0);
            memcpy(fixed.data() + n * rs, in + pos, rec);
            pos += rec; ++n;
            ends.push_back(pos);
        }
        int rc = g == 1 ? bn254_g1_decode_batch(ctx, fixed.data(), (bn_g1 *)out, status, n) : bn254_g2_decode_batch(ctx, fixed.data(), (bn_g2 *)out, status, n);
        if (rc) return rc;
        if (bn_opt(ctx, BN254_OPT_STREAM_STOP_AT_ERROR) == 1)      // the crate's own behaviour: Decodable returns Err at the first bad record
            for (size_t i = 0; i < n; ++i)
                if (status[i] != 0) { n = i + 1; pos = ends[i]; break; }
        *count = n; *consumed = pos;
        return BN254_OK;
    });
}
int bn254_g1_encode_stream(bn254_ctx *ctx, const bn_g1 *p, size_t n, uint8_t *out, size_t cap, size_t *written) { return stream_encode(ctx, 1, p, n, out, cap, written); }
int bn254_g2_encode_stream(bn254_ctx *ctx, const bn_g2 *p, size_t n, uint8_t *out, size_t cap, size_t *written) { return stream_encode(ctx, 2, p, n, out, cap, written); }
int bn254_g1_decode_stream(bn254_ctx *ctx, const uint8_t *in, size_t len, bn_g1 *out, int32_t *status, size_t max_points, size_t *count, size_t *consumed) {
    return stream_decode(ctx, 1, in, len, out, status, max_points, count, consumed);
}
int bn254_g2_decode_stream(bn254_ctx *ctx, const uint8_t *in, size_t len, bn_g2 *out, int32_t *status, size_t max_points, size_t *count, size_t *consumed) {
    return stream_decode(ctx, 2, in, len, out, status, max_points, count, consumed);
}
// G + G / G - G on host buffers
static int add_host(bn254_ctx *ctx, int g, const void *a, const void *b, void *out, size_t n, int negate_b) {
    if (n == 0) return BN254_OK;
    size_t ps = g == 1 ? sizeof(bn_g1) : sizeof(bn_g2);
    if (!a || !b || !out || n > 0x7fffffffu / 48) return BN254_E_BAD_ARG;
    BN_HOST_PROLOGUE();
    BnBuf &da = ctx->stage[0], &db = ctx->stage[1], &dout = ctx->stage[2];
    if ((rc = da.reserve(n * ps)) || (rc = db.reserve(n * ps)) || (rc = dout.reserve(n * ps))) return rc;
    HIP_TRY(hipMemcpyAsync(da.p, a, n * ps, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db.p, b, n * ps, hipMemcpyHostToDevice, ctx->stream));
    {
        BnScope sc(ctx, ctx->stream, g == 1 ? "g1_add" : "g2_add");
        rc = g == 1 ? bn254_launch_g1_add_M(da.p, db.p, dout.p, n, negate_b, ctx->stream) : bn254_launch_g2_add_M(da.p, db.p, dout.p, n, negate_b, ctx->stream);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(out, dout.p, n * ps, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BN254_OK;
}
int bn254_g1_add_batch(bn254_ctx *ctx, const bn_g1 *a, const bn_g1 *b, bn_g1 *out, size_t n, int negate_b) { return add_host(ctx, 1, a, b, out, n, negate_b); }
int bn254_g2_add_batch(bn254_ctx *ctx, const bn_g2 *a, const bn_g2 *b, bn_g2 *out, size_t n, int negate_b) { return add_host(ctx, 2, a, b, out, n, negate_b); }
// op 0: a * b, 1: a ^ k, 2: a^-1
static int gt_op_host(bn254_ctx *ctx, int op, const bn_gt *a, const void *b, size_t bsize, bn_gt *out, size_t n) {
    if (n == 0) return BN254_OK;
    if (!a || (op != 2 && !b) || !out || n > 0x7fffffffu / 96) return BN254_E_BAD_ARG;
    BN_HOST_PROLOGUE();
    BnBuf &da = ctx->stage[0], &db = ctx->stage[1], &dout = ctx->stage[2];
    if ((rc = da.reserve(n * sizeof(bn_gt))) || (rc = db.reserve(n * bsize)) || (rc = dout.reserve(n * sizeof(bn_gt)))) return rc;
    HIP_TRY(hipMemcpyAsync(da.p, a, n * sizeof(bn_gt), hipMemcpyHostToDevice, ctx->stream));
    if (op != 2) HIP_TRY(hipMemcpyAsync(db.p, b, n * bsize, hipMemcpyHostToDevice, ctx->stream));
    rc = op == 0 ? bn254_gt_mul_batch_dev(ctx, da.p, db.p, dout.p, n, ctx->stream)
       : op == 1 ? bn254_gt_pow_batch_dev(ctx, da.p, db.p, dout.p, n, ctx->stream)
                 : bn254_gt_inverse_batch_dev(ctx, da.p, dout.p, n, ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout.p, n * sizeof(bn_gt), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return BN254_OK;
}
int bn254_gt_mul_batch(bn254_ctx *ctx, const bn_gt *a, const bn_gt *b, bn_gt *out, size_t n) { return gt_op_host(ctx, 0, a, b, sizeof(bn_gt), out, n); }
int bn254_gt_pow_batch(bn254_ctx *ctx, const bn_gt *a, const bn_fr *k, bn_gt *out, size_t n) { return gt_op_host(ctx, 1, a, k, sizeof(bn_fr), out, n); }
int bn254_gt_inverse_batch(bn254_ctx *ctx, const bn_gt *a, bn_gt *out, size_t n) { return gt_op_host(ctx, 2, a, nullptr, 0, out, n); }

// ---------------------------------------------------------------------------------------------- measurement
int bn254_profile_enable(bn254_ctx *ctx, int on) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    ctx->profile = on != 0;
    return BN254_OK;
}
int bn254_profile_reset(bn254_ctx *ctx) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->prof_mu);
    for (auto &r : ctx->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    ctx->recs.clear();
    ctx->folded.clear();
    return BN254_OK;
}
// per-program time of the wave-cooperative machine on ONE wave (bn254_wave_ubench_W): ms for `iters` runs of program `which`
int bn254_wave_ubench(bn254_ctx *ctx, int which, int iters, double *ms_out) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (which < 0 || which > 5 || iters < 1 || !ms_out) return BN254_E_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    BnDeviceGuard dev_guard;
    HIP_TRY(hipSetDevice(ctx->device));
    BnBuf out;                                        // its own buffer: ctx->stage belongs to the host-buffer entry points
    if ((rc = out.reserve(4096))) return rc;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    auto body = [&]() -> int {
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        int r = bn254_launch_wave_ubench_W(which, 1, out.p, ctx->stream);           // warm-up (code and tables into the caches)
        if (r) return r;
        HIP_TRY(hipEventRecord(e0, ctx->stream));
        if ((r = bn254_launch_wave_ubench_W(which, iters, out.p, ctx->stream))) return r;
        HIP_TRY(hipEventRecord(e1, ctx->stream));
        HIP_TRY(hipEventSynchronize(e1));
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        return BN254_OK;
    };
    rc = body();
    if (rc) (void)hipStreamSynchronize(ctx->stream);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    out.release();
    if (!rc) *ms_out = ms;
    return rc;
}
int bn254_kernel_stats(bn254_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches) {
    int rc = bn_get_ctx(ctx); if (rc) return rc;
    if (!kernel || !total_ms || !launches) return BN254_E_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->prof_mu);
    fold_records(ctx, 0);                       // consumes the records: their events are destroyed here
    auto it = ctx->folded.find(kernel);
    *total_ms = it == ctx->folded.end() ? 0.0 : it->second.first;
    *launches = it == ctx->folded.end() ? 0 : it->second.second;
    return BN254_OK;
}

}  // extern "C"

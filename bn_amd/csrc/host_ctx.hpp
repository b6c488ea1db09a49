// Host-side internals shared by the translation units that implement include/bn254_hip.h (bn254_hip.hip: single-device
// entry points; bn254_multi.hip: pipelined host-buffer path, multi-device fan-out, RCCL exchange).  Not part of the ABI.
//
// Concurrency contract (what the header promises and these structures implement):
//   * any number of host threads may call the HOST-BUFFER entry points of one context - including the process-wide default
//     contexts behind ctx == NULL (one per HIP device) - and get the reference's re-entrant behaviour (`pairing` is a pure
//     function, `Group: Send + Sync`, src/lib.rs:55-61): the batch entry points (pairing_batch, g*_mul_batch) lease one of two
//     pipeline slots per call (own stream, staging and table; multi-chunk batches lease all slots), the others lock the context;
//   * the asynchronous *_dev entry points share context-owned scratch (final-exponentiation table, product workspace).
//     Each use is bracketed by an event: a launch on another stream first waits for the previous user's event, so two
//     streams on one context serialise on the scratch instead of racing on it.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <new>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bn254_hip.h"

#define HIP_TRY(expr)                             \
    do {                                          \
        hipError_t e__ = (expr);                  \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// restores the calling thread's current HIP device when an entry point returns (the library switches to the context's device)
struct BnDeviceGuard {
    int prev = -1;
    BnDeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~BnDeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};
// nothing may unwind across the C ABI: entry points that create threads / containers run their body through this
template <class Fn>
int bn_no_throw(Fn &&fn) {
    try { return fn(); } catch (const std::bad_alloc &) { return BN254_E_ALLOC; } catch (...) { return BN254_E_INTERNAL; }
}

#ifndef BN254_HAVE_QUAD
#define BN254_HAVE_QUAD 1        // the four-lanes-per-pairing kernels (bn254_kernels_q.hip) are linked in
#endif
constexpr int BN_MAX_SLOTS = 4;            // chunks in flight in the pipelined host-buffer path (2 used; the rest for experiments)

// grow-only buffer, device or pinned host
struct BnBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool pinned = false;
    int reserve(size_t need) {
        if (need == 0) need = 1;
        if (bytes >= need) return BN254_OK;
        release();
        hipError_t e = pinned ? hipHostMalloc(&p, need, hipHostMallocDefault) : hipMalloc(&p, need);
        if (e != hipSuccess) { p = nullptr; return BN254_E_ALLOC; }
        bytes = need;
        return BN254_OK;
    }
    void release() {
        if (p) { if (pinned) hipHostFree(p); else hipFree(p); }
        p = nullptr; bytes = 0;
    }
};

// one chunk in flight: its own stream, device staging and final-exponentiation table
struct BnSlot {
    hipStream_t stream = nullptr;
    BnBuf d_in[2], d_out, tbl;
};

struct bn254_ctx {
    int device = 0;
    int cus = 256;                      // compute units of the device (sizes one "round" of the lane-pair kernels: bn_round_pairs)
    std::atomic<long> opt[BN254_OPT_COUNT_];   // bn254_ctx_set_option: raw values, < 0 = "derive the default from the device" (bn_opt)
    std::mutex mu;                      // host-buffer entry points hold it for the whole call
    std::mutex scratch_mu;              // guards the scratch bookkeeping below (held only while enqueueing)
    hipStream_t stream = nullptr;       // the context's own stream (host-buffer entry points)
    BnBuf ws;                           // workspace (Miller values, product-tree levels)
    BnBuf exp_tbl;                      // odd-power tables of the windowed exponentiation by u (final_exp_B)
    BnBuf pow_tbl;                      // window tables of Gt::pow (gt_pow_B)
    BnBuf mul_tbl;                      // affine window tables of the scalar-multiplication kernels (one sub-launch)
    BnBuf miller_state;                 // running points of the shared-accumulator Miller loop (miller_shared*_B), one round
    hipEvent_t scratch_ev = nullptr;    // completion of the last launch that used ws / exp_tbl ...
    hipStream_t scratch_stream = nullptr;   // ... and the stream it ran on
    bool scratch_used = false;
    BnBuf stage[3];                     // device staging of the small host-buffer entry points
    BnSlot slot[BN_MAX_SLOTS];          // pipelined path (bn254_multi.hip)
    // leases of those slots: a batch of up to one chunk takes ONE of the first two (two callers overlap on the GPU - the number
    // of streams the hardware overlaps without loss), a multi-chunk batch takes all of them
    std::mutex slot_mu;
    std::condition_variable slot_cv;
    unsigned slot_busy = 0;             // bit i: slot i is leased
    int slot_waiting_all = 0;           // callers waiting for every slot (new single leases queue behind them)
    std::atomic<bool> profile{false};   // read by every launch helper, possibly from several host threads
    std::mutex prof_mu;                 // recs / folded (worker threads of the pipelined path launch concurrently)
    struct Rec { std::string name; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::map<std::string, std::pair<double, uint64_t>> folded;     // totals of records already consumed (events recycled)
};

// bn254_g2_prepare: the native line table of `nq` G2 points (bn254_kernels_b.hip NativeTableMem) and their infinity flags, in device memory
struct bn254_g2_prepared {
    int device = 0;
    size_t nq = 0;
    void *table = nullptr;
    void *inf = nullptr;
    size_t bytes = 0;
    // the points themselves (192 B each; ONE point: repeated `small_max` times), for calls small enough that the one-pairing-per-wave kernels of
    // the general path beat the lane-pair latency of the native Miller loop (bn254_pairing_prepared_native_batch_dev)
    void *q = nullptr;
    size_t small_max = 0;
};

// lease of pipeline slots for one host-buffer call (see bn254_ctx::slot_busy)
struct BnSlotLease {
    bn254_ctx *c; unsigned mask; int first;
    BnSlotLease(bn254_ctx *c_, bool all) : c(c_), mask(0), first(0) {
        std::unique_lock<std::mutex> lk(c->slot_mu);
        if (all) {
            ++c->slot_waiting_all;
            c->slot_cv.wait(lk, [&] { return c->slot_busy == 0; });
            --c->slot_waiting_all;
            mask = (1u << BN_MAX_SLOTS) - 1;
        } else {
            c->slot_cv.wait(lk, [&] { return c->slot_waiting_all == 0 && (c->slot_busy & 3u) != 3u; });
            first = (c->slot_busy & 1u) ? 1 : 0;
            mask = 1u << first;
        }
        c->slot_busy |= mask;
    }
    ~BnSlotLease() {
        { std::lock_guard<std::mutex> lk(c->slot_mu); c->slot_busy &= ~mask; }
        c->slot_cv.notify_all();
    }
};

// brackets one kernel launch with events when profiling is on
struct BnScope {
    bn254_ctx *c; hipStream_t s; bool on; hipEvent_t a, b; const char *name;
    BnScope(bn254_ctx *c_, hipStream_t s_, const char *n);
    ~BnScope();
};

int bn_get_ctx(bn254_ctx *&ctx);                                   // NULL -> the default context of the current device
long bn_opt(const bn254_ctx *c, int key);                          // EFFECTIVE value of a BN254_OPT_* tunable (defaults from the CU count)
int bn_debug_multi_exchange();                                     // BN254_EXCHANGE_AUTO unless the debug environment forces one (read once)
bool bn_debug_multi_affinity();                                    // false only when the debug environment says BN254_MULTI_AFFINITY=0
// scratch guard: lives across the enqueueing of work that reads/writes ctx->ws / ctx->exp_tbl on stream `s`
struct BnScratchGuard {
    bn254_ctx *c; hipStream_t s; int rc;
    BnScratchGuard(bn254_ctx *c_, hipStream_t s_);       // locks the bookkeeping, makes `s` wait for the previous user
    ~BnScratchGuard();                                   // records the completion event on `s`, unlocks
};

// kernel launch helpers (bn254_hip.hip); `table` = caller-provided final-exponentiation table or NULL for the context's own
size_t bn_round_pairs(const bn254_ctx *c);                        // pairings in one full-machine launch of the lane-pair kernels
size_t bn_sub_launch(const bn254_ctx *c, size_t n);                // sub-launch size for a batch of n (equal parts, none above one round)
int bn_launch_miller(bn254_ctx *c, const void *p, const void *q, void *f, size_t n, hipStream_t s, bool naf);
int bn_launch_pairing(bn254_ctx *c, const void *p, const void *q, void *out, size_t n, hipStream_t s, BnBuf *table);
int bn_launch_final_exp(bn254_ctx *c, const void *f, void *out, size_t n, hipStream_t s, BnBuf *table);
int bn_launch_product_final_exp(bn254_ctx *c, const void *in, size_t m, void *out, hipStream_t s);   // scratch guard held by the caller
int bn_launch_product(bn254_ctx *c, const void *in, size_t n, void *out, void *tmp, hipStream_t s);
size_t bn_product_tmp_bytes(const bn254_ctx *c, size_t n);
// table: the caller's own buffer (pipelined path: the slot's) or NULL for the context's (then under a BnScratchGuard)
int bn_mul_dev(bn254_ctx *ctx, int g, const void *d_p, const void *d_k, void *d_out, size_t n, hipStream_t s, int normalize, BnBuf *table = nullptr);

extern "C" {
// bn254_kernels_b.hip
int bn254_launch_miller_B(const void *p, const void *q, void *f, size_t n, int naf, hipStream_t s);
size_t bn254_miller_shared_state_bytes_B(size_t n, int m);
int bn254_launch_miller_shared_B(const void *p, const void *q, void *f, size_t n, int m, void *state, hipStream_t s);
int bn254_launch_final_exp_B(const void *f, void *out, size_t n, void *table, hipStream_t s);
size_t bn254_final_exp_table_bytes_B(size_t n);
int bn254_launch_g2_precompute_B(const void *q, void *coeffs, size_t n, hipStream_t s);
int bn254_launch_miller_prepared_B(const void *p, const void *coeffs, int shared, void *f, size_t n, hipStream_t s);
size_t bn254_native_table_bytes_B(size_t nq);
int bn254_native_lines_B(void);
int bn254_launch_g2_prepare_native_B(const void *q, void *table, void *q_inf, size_t nq, hipStream_t s);
int bn254_launch_miller_native_B(const void *p, const void *table, const void *q_inf, size_t nq, size_t q_lo, int shared, void *f, size_t n, hipStream_t s);
int bn254_launch_miller_native_shared_B(const void *p, const void *table, const void *q_inf, size_t nq, size_t q_lo, int shared, void *f, size_t n, int m, hipStream_t s);
int bn254_launch_gt_mul_B(const void *a, const void *b, void *out, size_t n, hipStream_t s);
size_t bn254_gt_pow_table_bytes_B(size_t n);
int bn254_launch_gt_pow_B(const void *a, const void *k, void *out, size_t n, void *table, int mode, hipStream_t s);
int bn254_launch_gt_inverse_B(const void *a, void *out, size_t n, hipStream_t s);
int bn254_launch_exp_by_neg_z_B(const void *a, void *out, size_t n, hipStream_t s);
// bn254_kernels_w.hip: one Fq12 per wave (wave.hpp)
int bn254_launch_wave_ubench_W(int which, int iters, void *out, hipStream_t s);
int bn254_launch_final_exp_W(const void *f, void *out, size_t n, hipStream_t s);
int bn254_launch_pairing_W(const void *p, const void *q, void *out, size_t n, int final_exp, hipStream_t s);
int bn254_launch_gt_tail_W(const void *in, size_t groups, unsigned m, void *out, int final_exp, hipStream_t s);
void bn254_gt_reduce_sizes_W(size_t n, unsigned chunk, unsigned per_wave, size_t *grid, size_t *scratch_bytes, size_t *counter_words);
int bn254_launch_gt_reduce_W(const void *in, size_t n, unsigned chunk, unsigned per_wave, unsigned bfly, void *scratch, void *counters, void *out, hipStream_t s);
// bn254_kernels_q.hip: one pairing per quad of lanes (quad.hpp)
int bn254_launch_miller_Q(const void *p, const void *q, void *f, size_t n, hipStream_t s);
size_t bn254_final_exp_table_bytes_Q(size_t n);
int bn254_launch_final_exp_Q(const void *f, void *out, size_t n, void *table, hipStream_t s);
// bn254_kernels_mul.hip
size_t bn254_mul_table_bytes_M(int g, size_t n);
int bn254_launch_g1_mul_M(const void *p, const void *k, void *out, size_t n, int normalize, void *table, hipStream_t s);
int bn254_launch_g2_mul_M(const void *p, const void *k, void *out, size_t n, int normalize, void *table, hipStream_t s);
int bn254_launch_g1_add_M(const void *a, const void *b, void *out, size_t n, int negate_b, hipStream_t s);
int bn254_launch_g2_add_M(const void *a, const void *b, void *out, size_t n, int negate_b, hipStream_t s);
}

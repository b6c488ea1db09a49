// Host orchestration of the BN254 engine around the kernels (include/bn254_hip.h):
//   * the pipelined host-buffer path: a batch handed over in pageable host memory is cut into chunks of 2^16; each of the two
//     chunks in flight has its own stream, device staging and exponentiation table, and a host thread that enqueues
//     H2D -> Miller -> final exponentiation -> D2H on that stream.  While one chunk is on the PCIe link the other owns the CUs;
//   * the multi-device fan-out of north_star: contiguous shards of independent pairings over the GPUs of one node (no
//     exchange), and the multi-pairing product: one un-exponentiated Fq12 per GPU, ONE RCCL all-gather of 384 bytes per rank
//     over xGMI (RCCL has no user-defined reduction), world-1 Fq12 products and a SINGLE final exponentiation - the fold of
//     the reference's shootout/main.rs:11-16, bit for bit, because the final exponentiation is a homomorphism;
// (The measurement and input-generation kernels of bench.py live in bn254_measure.hip.)
// RCCL is resolved with dlopen at the first multi-device call, so the library itself has no link-time dependency on it.
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

#include "host_ctx.hpp"
#include "bn254_constants.hpp"

// ============================================================================================ pipelined host-buffer path
// Measured on the box (profiles/r02b_stream_concurrency.txt, r02b_copy_rates.txt): hipMemcpy from/to PAGEABLE memory already runs at
// the link rate (51-53 GB/s; the runtime pins the pages on the fly), so no pinned staging copy is needed; and the GPU overlaps the
// pairing kernels of TWO streams perfectly (2 x 2^15 = 8.29 ms vs 8.40 ms in one launch) but loses with more (4 streams 13.2 ms,
// 8 streams 16.0 ms: the dispatcher stacks the small grids on the same SIMDs).  Hence: chunks of 2^16 (one full-machine launch),
// two chunks in flight on two streams, each driven by its own host thread which copies in, launches and copies out; while one chunk
// is on the PCIe link the other owns the CUs.
namespace {

struct MapJob {
    bn254_ctx *ctx;
    const char *in[2];
    size_t in_stride[2];
    char *out;
    size_t out_stride;
    size_t n, chunk;
    int nslots;
    std::function<int(BnSlot &, size_t)> launch;      // enqueue the kernels of one chunk on slot.stream (d_in -> d_out)
};

// worker `w` of `j.nslots` handles chunks w, w + nslots, ... on pipeline slot `slot_index`
int run_slot_chunks(const MapJob &j, int w, BnSlot &s) {
    for (size_t ci = (size_t)w; ci * j.chunk < j.n; ci += (size_t)j.nslots) {
        const size_t lo = ci * j.chunk, cnt = std::min(j.chunk, j.n - lo);
        int rc;
        for (int k = 0; k < 2; ++k) {
            const size_t b = cnt * j.in_stride[k];
            if ((rc = s.d_in[k].reserve(b))) return rc;
            HIP_TRY(hipMemcpyAsync(s.d_in[k].p, j.in[k] + lo * j.in_stride[k], b, hipMemcpyHostToDevice, s.stream));
        }
        const size_t ob = cnt * j.out_stride;
        if ((rc = s.d_out.reserve(ob))) return rc;
        if ((rc = j.launch(s, cnt))) return rc;
        HIP_TRY(hipMemcpyAsync(j.out + lo * j.out_stride, s.d_out.p, ob, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipStreamSynchronize(s.stream));
    }
    return BN254_OK;
}
int run_slot(const MapJob &j, int w, int slot_index) {
    bn254_ctx *c = j.ctx;
    BnSlot &s = c->slot[slot_index];
    HIP_TRY(hipSetDevice(c->device));
    if (!s.stream) HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    const int rc = run_slot_chunks(j, w, s);
    // On an error return copies that read or write the CALLER's buffers may still be in flight on this stream: drain it before
    // the C call returns and the slot lease passes to the next caller.
    if (rc) (void)hipStreamSynchronize(s.stream);
    return rc;
}

void plan(const bn254_ctx *c, size_t n, size_t &chunk, int &nslots) {
    // as few chunks as possible with none above one round of the machine (256 pairings per CU: two waves on every SIMD), all of
    // (nearly) the same size: a ragged tail of a few pairings would cost a whole kernel latency
    size_t ch = bn_sub_launch(c, n);
    const long forced = bn_opt(c, BN254_OPT_PIPELINE_CHUNK);                       // experiments
    if (forced > 0) ch = (size_t)forced;
    chunk = ch;
    const int want = std::min(BN_MAX_SLOTS, std::max(1, (int)bn_opt(c, BN254_OPT_PIPELINE_SLOTS)));
    nslots = (int)std::min<size_t>((size_t)want, (n + ch - 1) / ch);
}

// runs fn(w) for w = 0 .. count-1, workers 1.. on their own host threads.  Nothing may unwind past a joinable std::thread (its
// destructor calls std::terminate): a thread constructor that throws (resource exhaustion) makes the remaining workers run inline, and
// an exception out of fn - bad_alloc inside a std::function or vector, say - on whichever thread is held until every started thread
// has been joined, then rethrown on the calling thread for the entry point's bn_no_throw to translate.
// all_on_workers: job 0 gets a thread of its own too, so that nothing a job does to its thread (BnAffinityScope pins it) touches the CALLER's
// thread - threads the HIP / RCCL runtime creates lazily from a pinned thread inherit the narrowed mask for good
template <class Fn>
void run_workers(int count, Fn fn, bool all_on_workers = false) {
    std::vector<std::exception_ptr> err((size_t)count);
    auto guarded = [&](int w) noexcept { try { fn(w); } catch (...) { err[(size_t)w] = std::current_exception(); } };
    std::vector<std::thread> th;
    th.reserve((size_t)count);
    const int first = all_on_workers ? 0 : 1;
    int started = first;
    for (int w = first; w < count; ++w) {
        try { th.emplace_back(guarded, w); ++started; } catch (...) { break; }
    }
    if (!all_on_workers) guarded(0);
    for (int w = started; w < count; ++w) guarded(w);                    // (thread creation failed: the rest runs here, one after the other)
    for (auto &t : th) t.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
}

int run_map(MapJob &j) {
    plan(j.ctx, j.n, j.chunk, j.nslots);
    if (j.nslots <= 1) {                                      // one chunk: one slot, on the calling thread; a second caller overlaps
        BnSlotLease lease(j.ctx, false);
        return run_slot(j, 0, lease.first);
    }
    BnSlotLease lease(j.ctx, true);
    std::vector<int> rcs(j.nslots, BN254_OK);
    run_workers(j.nslots, [&](int w) { rcs[w] = run_slot(j, w, w); });
    for (int rc : rcs) if (rc) return rc;
    return BN254_OK;
}

}  // namespace

// thread-safe by slot leases (no context mutex): up to two single-chunk callers run concurrently
int bn_pairing_batch_pipelined(bn254_ctx *ctx, const bn_g1 *p, const bn_g2 *q, bn_gt *out, size_t n) {
    MapJob j;
    j.ctx = ctx; j.n = n;
    j.in[0] = (const char *)p; j.in_stride[0] = sizeof(bn_g1);
    j.in[1] = (const char *)q; j.in_stride[1] = sizeof(bn_g2);
    j.out = (char *)out; j.out_stride = sizeof(bn_gt);
    j.launch = [ctx](BnSlot &s, size_t cnt) {
        // (large chunks: Miller values land in the output slots and are exponentiated in place; the table is the slot's own)
        return bn_launch_pairing(ctx, s.d_in[0].p, s.d_in[1].p, s.d_out.p, cnt, s.stream, &s.tbl);
    };
    return run_map(j);
}
int bn_mul_batch_pipelined(bn254_ctx *ctx, int g, const void *p, const bn_fr *k, void *out, size_t n) {
    const size_t ps = g == 1 ? sizeof(bn_g1) : sizeof(bn_g2);
    MapJob j;
    j.ctx = ctx; j.n = n;
    j.in[0] = (const char *)p; j.in_stride[0] = ps;
    j.in[1] = (const char *)k; j.in_stride[1] = sizeof(bn_fr);
    j.out = (char *)out; j.out_stride = ps;
    j.launch = [ctx, g](BnSlot &s, size_t cnt) { return bn_mul_dev(ctx, g, s.d_in[0].p, s.d_in[1].p, s.d_out.p, cnt, s.stream, 1, &s.tbl); };
    return run_map(j);
}

// ============================================================================================ multi-device
namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
std::mutex g_rccl_mu;
Rccl g_rccl;

Rccl &rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h || g_rccl.ok) return g_rccl;
    // a process that already runs on PyTorch's bundled HIP runtime must use PyTorch's bundled RCCL (bn_amd/_native.py sets
    // BN254_RCCL_PATH); a plain C/C++/Rust host gets /opt/rocm's
    const char *cands[] = {getenv("BN254_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *c : cands) {
        if (!c || !*c) continue;
        g_rccl.h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) return g_rccl;
    auto sym = [&](const char *n) { return dlsym(g_rccl.h, n); };
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))sym("ncclCommAbort");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.ok = g_rccl.CommInitAll && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.GroupStart && g_rccl.GroupEnd;
    return g_rccl;
}

}  // namespace

// ---- host-thread placement: the thread that drives a rank (pageable H2D / D2H copies, launches) runs on the CPUs of that GPU's NUMA
// node when the node is known (/sys/bus/pci/devices/<bus id>/numa_node) and the process may use some of its CPUs; otherwise it is
// left alone.  Eight ranks driven from one process otherwise all stage their copies from whatever node the caller happens to run on.
struct RankCpus { int node = -1; cpu_set_t set; bool valid = false; };
static RankCpus rank_cpus_of_device(int dev) {
    RankCpus r; CPU_ZERO(&r.set);
    char bus[64] = {};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) return r;
    for (char *c = bus; *c; ++c) *c = (char)tolower(*c);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return r;
    int node = -1;
    const int got = fscanf(f, "%d", &node); fclose(f);
    if (got != 1 || node < 0) return r;
    r.node = node;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    if (!(f = fopen(path, "r"))) return r;
    char list[4096] = {};
    const bool have = fgets(list, sizeof list, f) != nullptr; fclose(f);
    if (!have) return r;
    char *save = nullptr;
    for (char *tok = strtok_r(list, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {     // "0-15,128-143"
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k < 1) continue;
        if (k == 1) b = a;
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) if (c >= 0) CPU_SET(c, &r.set);
    }
    cpu_set_t cur;
    if (sched_getaffinity(0, sizeof cur, &cur) == 0) CPU_AND(&r.set, &r.set, &cur);
    r.valid = CPU_COUNT(&r.set) > 0;
    return r;
}
// pins the CURRENT thread for the lifetime of the object - worker threads only (they end with the call): the caller's thread is never
// re-pinned, every rank of a multi-device call runs on a thread of its own (run_workers all_on_workers)
struct BnAffinityScope {
    cpu_set_t old; bool restore = false;
    BnAffinityScope(const RankCpus &rc, std::thread::id caller) {
        if (!rc.valid || std::this_thread::get_id() == caller) return;
        if (pthread_getaffinity_np(pthread_self(), sizeof old, &old) != 0) return;
        restore = pthread_setaffinity_np(pthread_self(), sizeof rc.set, &rc.set) == 0;
    }
    ~BnAffinityScope() { if (restore) pthread_setaffinity_np(pthread_self(), sizeof old, &old); }
};

struct bn254_multi {
    std::vector<RankCpus> cpus;             // per rank: the CPUs of its GPU's NUMA node (valid = false: no pinning)
    std::vector<int> devices;
    std::vector<bn254_ctx *> ctx;
    std::vector<ncclComm_t> comms;          // one per rank when the exchange is RCCL
    std::vector<BnBuf> d_partial, d_gather; // 384 B / world x 384 B per rank, on that rank's device
    int exchange = BN254_EXCHANGE_PEER;
    std::mutex mu;                          // one multi-device call at a time per handle
};

// bn254_g2_prepare_multi: one native prepared handle per rank (the whole point for nq == 1, the rank's shard of the points otherwise)
struct bn254_multi_prepared {
    bn254_multi *owner = nullptr;
    size_t nq = 0;
    std::vector<bn254_g2_prepared *> h;
};

extern "C" {

static int multi_create(const int *devices, int ndev, int exchange, bn254_multi **out);
static int pairing_batch_multi(bn254_multi *m, const bn_g1 *p, const bn_g2 *q, bn_gt *out, size_t n);
static int pairing_product_multi(bn254_multi *m, const bn_g1 *p, const bn_g2 *q, size_t n, bn_gt *out);
int bn254_multi_create_ex(const int *devices, int ndev, int exchange, bn254_multi **out) {
    if (exchange != BN254_EXCHANGE_AUTO && exchange != BN254_EXCHANGE_PEER && exchange != BN254_EXCHANGE_RCCL) return BN254_E_BAD_ARG;
    BnDeviceGuard dev_guard;
    return bn_no_throw([&] { return multi_create(devices, ndev, exchange, out); });
}
int bn254_multi_create(const int *devices, int ndev, bn254_multi **out) {
    return bn254_multi_create_ex(devices, ndev, bn_debug_multi_exchange(), out);          // AUTO unless the debug environment says otherwise
}
int bn254_multi_set_option(bn254_multi *m, int key, long value) {
    if (!m) return BN254_E_BAD_ARG;
    for (int g = 0; g < bn254_multi_device_count(m); ++g) {
        const int rc = bn254_ctx_set_option(bn254_multi_ctx(m, g), key, value);
        if (rc) return rc;
    }
    return BN254_OK;
}
int bn254_pairing_batch_multi(bn254_multi *m, const bn_g1 *p, const bn_g2 *q, bn_gt *out, size_t n) {
    BnDeviceGuard dev_guard;
    return bn_no_throw([&] { return pairing_batch_multi(m, p, q, out, n); });
}
int bn254_pairing_product_multi(bn254_multi *m, const bn_g1 *p, const bn_g2 *q, size_t n, bn_gt *out) {
    BnDeviceGuard dev_guard;
    return bn_no_throw([&] { return pairing_product_multi(m, p, q, n, out); });
}
static int multi_create(const int *devices, int ndev, int exchange, bn254_multi **out) {
    if (!out || ndev <= 0 || ndev > 64) return BN254_E_BAD_ARG;
    const int have = bn254_device_count();
    if (have <= 0) return BN254_E_NO_DEVICE;
    bn254_multi *m = new bn254_multi();
    for (int g = 0; g < ndev; ++g) {
        int d = devices ? devices[g] : g;
        if (d < 0 || d >= have) { bn254_multi_destroy(m); return BN254_E_NO_DEVICE; }
        bn254_ctx *c = nullptr;
        int rc = bn254_ctx_create(d, &c);
        if (rc) { bn254_multi_destroy(m); return rc; }
        m->devices.push_back(d);
        m->ctx.push_back(c);
        m->cpus.push_back(bn_debug_multi_affinity() ? rank_cpus_of_device(d) : RankCpus{});
    }
    m->d_partial.resize(ndev); m->d_gather.resize(ndev);
    for (int g = 0; g < ndev; ++g) {
        if (hipSetDevice(m->devices[g]) != hipSuccess) { bn254_multi_destroy(m); return BN254_E_NO_DEVICE; }
        int rc;
        if ((rc = m->d_partial[g].reserve(sizeof(bn_gt))) || (rc = m->d_gather[g].reserve((size_t)ndev * sizeof(bn_gt)))) { bn254_multi_destroy(m); return rc; }
    }
    // RCCL needs one distinct GPU per rank; a device list with repeats (several contexts on one GPU: how the N > 1 control
    // flow is exercised on a one-GPU box) exchanges the partials with peer copies instead
    std::vector<int> sorted = m->devices;
    std::sort(sorted.begin(), sorted.end());
    const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
    const bool want_rccl = distinct && exchange != BN254_EXCHANGE_PEER;
    if (want_rccl) {
        Rccl &r = rccl();
        if (r.ok) {
            m->comms.resize(ndev);
            if (r.CommInitAll(m->comms.data(), ndev, m->devices.data()) == ncclSuccess) m->exchange = BN254_EXCHANGE_RCCL;
            else m->comms.clear();
        }
    }
    if (m->exchange != BN254_EXCHANGE_RCCL && exchange == BN254_EXCHANGE_RCCL) { bn254_multi_destroy(m); return BN254_E_COMM; }
    *out = m;
    return BN254_OK;
}
void bn254_multi_destroy(bn254_multi *m) {
    if (!m) return;
    BnDeviceGuard dev_guard;
    if (!m->comms.empty()) for (auto c : m->comms) if (c) rccl().CommDestroy(c);
    for (size_t g = 0; g < m->devices.size(); ++g) {
        hipSetDevice(m->devices[g]);
        if (g < m->d_partial.size()) m->d_partial[g].release();
        if (g < m->d_gather.size()) m->d_gather[g].release();
    }
    for (auto c : m->ctx) bn254_ctx_destroy(c);
    delete m;
}
int bn254_multi_device_count(const bn254_multi *m) { return m ? (int)m->devices.size() : 0; }
int bn254_multi_rank_numa_node(const bn254_multi *m, int rank) {
    return (m && rank >= 0 && rank < (int)m->cpus.size() && m->cpus[(size_t)rank].valid) ? m->cpus[(size_t)rank].node : -1;
}
int bn254_multi_exchange_kind(const bn254_multi *m) { return m ? m->exchange : BN254_E_BAD_ARG; }
bn254_ctx *bn254_multi_ctx(bn254_multi *m, int rank) { return (m && rank >= 0 && rank < (int)m->ctx.size()) ? m->ctx[rank] : nullptr; }

// contiguous shards lo = n*g/G .. n*(g+1)/G (the rule of bn_amd.distributed.shard_range); no exchange
static int pairing_batch_multi(bn254_multi *m, const bn_g1 *p, const bn_g2 *q, bn_gt *out, size_t n) {
    if (!m) return BN254_E_BAD_ARG;
    if (n == 0) return BN254_OK;
    if (!p || !q || !out) return BN254_E_BAD_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    const size_t G = m->ctx.size();
    std::vector<int> rcs(G, BN254_OK);
    const std::thread::id caller = std::this_thread::get_id();
    run_workers((int)G, [&](int g) {
        BnAffinityScope pin(m->cpus[(size_t)g], caller);
        const size_t lo = n * (size_t)g / G, hi = n * ((size_t)g + 1) / G;
        rcs[g] = bn254_pairing_batch(m->ctx[g], p + lo, q + lo, out + lo, hi - lo);
    }, true);
    for (int rc : rcs) if (rc) return rc;
    return BN254_OK;
}

void bn254_multi_prepared_destroy(bn254_multi_prepared *prep) {
    if (!prep) return;
    for (auto h : prep->h) bn254_g2_prepared_destroy(h);
    delete prep;
}
size_t bn254_multi_prepared_count(const bn254_multi_prepared *prep) { return prep ? prep->nq : 0; }
static int g2_prepare_multi(bn254_multi *m, const bn_g2 *q, size_t nq, bn254_multi_prepared **out) {
    if (!out) return BN254_E_BAD_ARG;
    *out = nullptr;
    if (!m || !q || nq == 0) return BN254_E_BAD_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    const size_t G = m->ctx.size();
    if (nq != 1 && nq < G) return BN254_E_BAD_ARG;                       // every rank needs at least one point of a sharded set
    bn254_multi_prepared *pr = new bn254_multi_prepared();
    pr->owner = m; pr->nq = nq; pr->h.assign(G, nullptr);
    std::vector<int> rcs(G, BN254_OK);
    const std::thread::id caller = std::this_thread::get_id();
    run_workers((int)G, [&](int g) {
        BnAffinityScope pin(m->cpus[(size_t)g], caller);
        const size_t lo = nq == 1 ? 0 : nq * (size_t)g / G, hi = nq == 1 ? 1 : nq * ((size_t)g + 1) / G;
        rcs[g] = bn254_g2_prepare(m->ctx[g], q + lo, hi - lo, &pr->h[(size_t)g]);
    }, true);
    for (int rc : rcs) if (rc) { bn254_multi_prepared_destroy(pr); return rc; }
    *out = pr;
    return BN254_OK;
}
static int pairing_prepared_native_batch_multi(bn254_multi *m, const bn_g1 *p, const bn254_multi_prepared *prep, bn_gt *out, size_t n) {
    if (!m || !prep || prep->owner != m) return BN254_E_BAD_ARG;
    if (n == 0) return BN254_OK;
    if (!p || !out || (prep->nq != 1 && n != prep->nq)) return BN254_E_BAD_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    const size_t G = m->ctx.size();
    std::vector<int> rcs(G, BN254_OK);
    const std::thread::id caller = std::this_thread::get_id();
    run_workers((int)G, [&](int g) {
        BnAffinityScope pin(m->cpus[(size_t)g], caller);
        const size_t lo = n * (size_t)g / G, hi = n * ((size_t)g + 1) / G;               // the rule the points were sharded by
        rcs[g] = bn254_pairing_prepared_native_batch(m->ctx[g], p + lo, prep->h[(size_t)g], out + lo, hi - lo);
    }, true);
    for (int rc : rcs) if (rc) return rc;
    return BN254_OK;
}
int bn254_g2_prepare_multi(bn254_multi *m, const bn_g2 *q, size_t nq, bn254_multi_prepared **out) {
    BnDeviceGuard dev_guard;
    return bn_no_throw([&] { return g2_prepare_multi(m, q, nq, out); });
}
int bn254_pairing_prepared_native_batch_multi(bn254_multi *m, const bn_g1 *p, const bn254_multi_prepared *prep, bn_gt *out, size_t n) {
    BnDeviceGuard dev_guard;
    return bn_no_throw([&] { return pairing_prepared_native_batch_multi(m, p, prep, out, n); });
}

// q != NULL: the fused multi-pairing of (p[i], q[i]);  prep != NULL: the same over natively prepared points (one point on every rank, or the set
// sharded by the rule of the pairs - then n == count)
static int product_multi_impl(bn254_multi *m, const bn_g1 *p, const bn_g2 *q, const bn254_multi_prepared *prep, size_t n, bn_gt *out) {
    if (!m || !out || (n && (!p || (!q && !prep)))) return BN254_E_BAD_ARG;
    if (prep && (prep->owner != m || (prep->nq != 1 && n != prep->nq))) return BN254_E_BAD_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    const size_t G = m->ctx.size();
    std::vector<int> rcs(G, BN254_OK);
    // 1. every rank: its shard's Miller loops and Fq12 product tree -> ONE un-exponentiated Fq12 in d_partial[g]
    auto local = [&](size_t g) -> int {
        bn254_ctx *c = m->ctx[g];
        const size_t lo = n * g / G, cnt = n * (g + 1) / G - lo;
        std::lock_guard<std::mutex> cl(c->mu);
        HIP_TRY(hipSetDevice(c->device));
        BnBuf &dp = c->stage[0], &dq = c->stage[1];
        int rc;
        if ((rc = dp.reserve(cnt * sizeof(bn_g1))) || (!prep && (rc = dq.reserve(cnt * sizeof(bn_g2))))) return rc;
        if (cnt) {
            HIP_TRY(hipMemcpyAsync(dp.p, p + lo, cnt * sizeof(bn_g1), hipMemcpyHostToDevice, c->stream));
            if (!prep) HIP_TRY(hipMemcpyAsync(dq.p, q + lo, cnt * sizeof(bn_g2), hipMemcpyHostToDevice, c->stream));
        }
        if (prep) rc = bn254_miller_product_prepared_native_dev(c, dp.p, prep->h[g], 0, cnt, m->d_partial[g].p, c->stream);
        else rc = bn254_miller_product_dev(c, dp.p, dq.p, cnt, m->d_partial[g].p, c->stream);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        return BN254_OK;
    };
    const std::thread::id caller = std::this_thread::get_id();
    run_workers((int)G, [&](int g) { BnAffinityScope pin(m->cpus[(size_t)g], caller); rcs[g] = local((size_t)g); }, true);
    for (int rc : rcs) if (rc) return rc;
    // 2. the ONE exchange step: 384 bytes per rank
    if (m->exchange == BN254_EXCHANGE_RCCL) {
        Rccl &r = rccl();
        // A failed collective leaves work enqueued on the earlier ranks' streams and the communicators unusable: abort them (a later
        // call or ncclCommDestroy on a wedged communicator can hang), fall back to the peer-copy exchange for the following calls,
        // and report BN254_E_COMM for this one.
        auto fail = [&]() -> int {
            for (auto &c : m->comms) if (c) { if (r.CommAbort) r.CommAbort(c); else r.CommDestroy(c); c = nullptr; }
            m->comms.clear();
            m->exchange = BN254_EXCHANGE_PEER;
            for (size_t g = 0; g < G; ++g) { hipSetDevice(m->devices[g]); (void)hipStreamSynchronize(m->ctx[g]->stream); }
            return BN254_E_COMM;
        };
        if (r.GroupStart() != ncclSuccess) return fail();
        bool ok = true;
        for (size_t g = 0; g < G && ok; ++g) {
            hipSetDevice(m->devices[g]);
            ok = r.AllGather(m->d_partial[g].p, m->d_gather[g].p, 48, ncclUint64, m->comms[g], m->ctx[g]->stream) == ncclSuccess;
        }
        if (r.GroupEnd() != ncclSuccess || !ok) return fail();
        for (size_t g = 1; g < G; ++g) { HIP_TRY(hipSetDevice(m->devices[g])); HIP_TRY(hipStreamSynchronize(m->ctx[g]->stream)); }
    } else {
        HIP_TRY(hipSetDevice(m->devices[0]));
        for (size_t g = 0; g < G; ++g) {
            char *dst = (char *)m->d_gather[0].p + g * sizeof(bn_gt);
            if (m->devices[g] == m->devices[0]) HIP_TRY(hipMemcpyAsync(dst, m->d_partial[g].p, sizeof(bn_gt), hipMemcpyDeviceToDevice, m->ctx[0]->stream));
            else HIP_TRY(hipMemcpyPeerAsync(dst, m->devices[0], m->d_partial[g].p, m->devices[g], sizeof(bn_gt), m->ctx[0]->stream));
        }
    }
    // 3. rank 0: world-1 multiplications and the single final exponentiation
    bn254_ctx *c0 = m->ctx[0];
    std::lock_guard<std::mutex> cl(c0->mu);
    HIP_TRY(hipSetDevice(c0->device));
    int rc;
    if ((rc = bn254_gt_product_final_exp_dev(c0, m->d_gather[0].p, G, m->d_partial[0].p, c0->stream))) return rc;      // ONE launch
    HIP_TRY(hipMemcpyAsync(out, m->d_partial[0].p, sizeof(bn_gt), hipMemcpyDeviceToHost, c0->stream));
    HIP_TRY(hipStreamSynchronize(c0->stream));
    return BN254_OK;
}
static int pairing_product_multi(bn254_multi *m, const bn_g1 *p, const bn_g2 *q, size_t n, bn_gt *out) {
    if (n && !q) return BN254_E_BAD_ARG;
    return product_multi_impl(m, p, q, nullptr, n, out);
}
int bn254_pairing_product_prepared_native_multi(bn254_multi *m, const bn_g1 *p, const bn254_multi_prepared *prep, size_t n, bn_gt *out) {
    if (!prep) return BN254_E_BAD_ARG;
    BnDeviceGuard dev_guard;
    return bn_no_throw([&] { return product_multi_impl(m, p, nullptr, prep, n, out); });
}

}  // extern "C"

// PCIe-inclusive rate of the C ABI from a compiled host with no Python in the process (what a Rust binding of the reference's
// `pairing` sees): n random (P, Q) in pageable std::vector memory -> bn254_pairing_batch -> n Gt in pageable memory.
//   g++ -O2 -std=c++17 -Iinclude tools/host_api_bench.cpp -Lbn_amd -lbn254_hip -Wl,-rpath,$PWD/bn_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/host_api_bench
//   /tmp/host_api_bench [n] [threads]        threads > 1: that many host threads call bn254_pairing_batch(NULL, ...) concurrently
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "bn254.hpp"

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : (size_t)1 << 16;
    const int threads = argc > 2 ? atoi(argv[2]) : 1;
    std::mt19937_64 rng(42);
    // random scalars as 32-byte big-endian integers below 2^248 (< r), decoded to the crate's Montgomery image by the library
    std::vector<uint8_t> bytes(2 * n * 32);
    for (size_t i = 0; i < 2 * n; ++i) { bytes[32 * i] = 0; for (int j = 1; j < 32; ++j) bytes[32 * i + j] = (uint8_t)rng(); }
    std::vector<bn_fr> k(2 * n);
    std::vector<int32_t> st(2 * n);
    bn::check(bn254_fr_decode_batch(nullptr, bytes.data(), k.data(), st.data(), 2 * n));
    std::vector<bn::G1> g1(n, bn::G1::one()), p(n);
    std::vector<bn::G2> g2(n, bn::G2::one()), q(n);
    bn::check(bn254_g1_mul_batch(nullptr, (const bn_g1 *)g1.data(), k.data(), (bn_g1 *)p.data(), n));
    bn::check(bn254_g2_mul_batch(nullptr, (const bn_g2 *)g2.data(), k.data() + n, (bn_g2 *)q.data(), n));
    std::vector<bn::Gt> out(n, bn::Gt::one());                       // initialised, like a Rust vec![Gt::one(); n]: pages resident
    bn::check(bn254_pairing_batch(nullptr, (const bn_g1 *)p.data(), (const bn_g2 *)q.data(), (bn_gt *)out.data(), n));   // warm-up: allocations
    const int reps = 5;
    auto t0 = std::chrono::steady_clock::now();
    if (threads <= 1) {
        for (int r = 0; r < reps; ++r) bn::check(bn254_pairing_batch(nullptr, (const bn_g1 *)p.data(), (const bn_g2 *)q.data(), (bn_gt *)out.data(), n));
    } else {
        std::vector<std::thread> th;
        const size_t per = n / threads;
        for (int t = 0; t < threads; ++t)
            th.emplace_back([&, t] {
                for (int r = 0; r < reps; ++r)
                    bn::check(bn254_pairing_batch(nullptr, (const bn_g1 *)p.data() + t * per, (const bn_g2 *)q.data() + t * per, (bn_gt *)out.data() + t * per, per));
            });
        for (auto &x : th) x.join();
    }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    // the whole (possibly multi-threaded) result against one more single-threaded pass; not constant, not one
    std::vector<bn::Gt> again(n);
    bn::check(bn254_pairing_batch(nullptr, (const bn_g1 *)p.data(), (const bn_g2 *)q.data(), (bn_gt *)again.data(), n));
    bool ok = !(again[0] == bn::Gt::one()) && !(again[0] == again[1]);
    const size_t covered = threads <= 1 ? n : (n / threads) * threads;
    for (size_t i = 0; i < covered; ++i) ok = ok && again[i] == out[i];
    std::printf("C++ host, bn254_pairing_batch(NULL ctx), n = %zu, %d calling thread(s): %.2f ms per pass = %.3f M pairings/s (PCIe included)%s\n",
                n, threads, dt * 1e3, n / dt / 1e6, ok ? "" : "  RESULT CHECK FAILED");
    return ok ? 0 : 1;
}

// bn254.hpp - header-only C++ facade over the C ABI (bn254_hip.h), mirroring the public API of the reference crate
// zcash-hackworks/bn (src/lib.rs): same names, argument meaning and error behaviour, so code written against the crate reads
// the same here.  Values are the crate's #[repr(C)] memory images; all curve/pairing arithmetic runs on the GPU.
//
//   reference (Rust)                              here (C++)
//   bn::pairing(p, q) -> Gt        lib.rs:181     bn::pairing(p, q)
//   G1::one() / zero() / is_zero   lib.rs:83-87   bn::G1::one() / zero() / is_zero()
//   g * fr                         lib.rs:116     g * fr                  (returned normalized, lib.rs:88-95)
//   Gt::one(), a == b              lib.rs:169     bn::Gt::one(), a == b   (canonical limbs: memcmp)
//   (fold of shootout/main.rs)                    bn::pairing_batch(...), bn::pairing_product(...)
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "bn254_hip.h"

namespace bn {

struct Error : std::runtime_error {
    int code;
    explicit Error(int c) : std::runtime_error(std::string("bn254_hip: ") + bn254_error_string(c)), code(c) {}
};
inline void check(int rc) { if (rc != 0) throw Error(rc); }

// R mod q, the Montgomery image of 1 (fp.rs:170-177)
static const uint64_t FQ_ONE[4] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full};
static const uint64_t FQ_TWO[4] = {0xa6ba871b8b1e1b3aull, 0x14f1d651eb8e167bull, 0xccdd46def0f28c58ull, 0x1c14ef83340fbe5eull};
static const uint64_t FR_ONE[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};

struct Fr {
    bn_fr v;
    static Fr one() { Fr r; std::memcpy(r.v.l, FR_ONE, 32); return r; }     // lib.rs:21
    static Fr zero() { Fr r; std::memset(&r.v, 0, 32); return r; }          // lib.rs:20
};
struct G1 {
    bn_g1 v;
    static G1 one() {                                                        // groups/mod.rs:355-361
        G1 r; std::memcpy(r.v.x, FQ_ONE, 32); std::memcpy(r.v.y, FQ_TWO, 32); std::memcpy(r.v.z, FQ_ONE, 32); return r;
    }
    static G1 zero() { G1 r; std::memset(&r.v, 0, sizeof r.v); std::memcpy(r.v.y, FQ_ONE, 32); return r; }   // (0,1,0)
    bool is_zero() const { return (v.z[0] | v.z[1] | v.z[2] | v.z[3]) == 0; }
    G1 operator*(const Fr &k) const { G1 r; check(bn254_g1_mul_batch(nullptr, &v, &k.v, &r.v, 1)); return r; }
    G1 operator+(const G1 &o) const { G1 r; check(bn254_g1_add_batch(nullptr, &v, &o.v, &r.v, 1, 0)); return r; }     // lib.rs:103-106
    G1 operator-(const G1 &o) const { G1 r; check(bn254_g1_add_batch(nullptr, &v, &o.v, &r.v, 1, 1)); return r; }     // lib.rs:108-111
    G1 operator-() const { return zero() - *this; }                                                                  // lib.rs:113-114
    void normalize() { *this = *this * Fr::one(); }                          // lib.rs:88-95
};
struct G2 {
    bn_g2 v;
    static G2 one() {                                                        // groups/mod.rs:377-390
        static const uint64_t X[8] = {0x8e83b5d102bc2026ull, 0xdceb1935497b0172ull, 0xfbb8264797811adfull, 0x19573841af96503bull,
                                      0xafb4737da84c6140ull, 0x6043dd5a5802d8c4ull, 0x09e950fc52a02f86ull, 0x14fef0833aea7b6bull};
        static const uint64_t Y[8] = {0x619dfa9d886be9f6ull, 0xfe7fd297f59e9b78ull, 0xff9e1a62231b7dfeull, 0x28fd7eebae9e4206ull,
                                      0x64095b56c71856eeull, 0xdc57f922327d3cbbull, 0x55f935be33351076ull, 0x0da4a0e693fd6482ull};
        G2 r; std::memcpy(r.v.x, X, 64); std::memcpy(r.v.y, Y, 64); std::memset(r.v.z, 0, 64); std::memcpy(r.v.z, FQ_ONE, 32); return r;
    }
    static G2 zero() { G2 r; std::memset(&r.v, 0, sizeof r.v); std::memcpy(r.v.y, FQ_ONE, 32); return r; }
    bool is_zero() const { uint64_t o = 0; for (int i = 0; i < 8; ++i) o |= v.z[i]; return o == 0; }
    G2 operator*(const Fr &k) const { G2 r; check(bn254_g2_mul_batch(nullptr, &v, &k.v, &r.v, 1)); return r; }
    G2 operator+(const G2 &o) const { G2 r; check(bn254_g2_add_batch(nullptr, &v, &o.v, &r.v, 1, 0)); return r; }
    G2 operator-(const G2 &o) const { G2 r; check(bn254_g2_add_batch(nullptr, &v, &o.v, &r.v, 1, 1)); return r; }
    G2 operator-() const { return zero() - *this; }
    void normalize() { *this = *this * Fr::one(); }
};
struct Gt {
    bn_gt v;
    static Gt one() { Gt r; std::memset(&r.v, 0, sizeof r.v); std::memcpy(r.v.c, FQ_ONE, 32); return r; }     // lib.rs:169
    Gt operator*(const Gt &o) const { Gt r; check(bn254_gt_mul_batch(nullptr, &v, &o.v, &r.v, 1)); return r; }      // lib.rs:175-179
    Gt pow(const Fr &k) const { Gt r; check(bn254_gt_pow_batch(nullptr, &v, &k.v, &r.v, 1)); return r; }            // lib.rs:171
    Gt inverse() const { Gt r; check(bn254_gt_inverse_batch(nullptr, &v, &r.v, 1)); return r; }                     // lib.rs:172
    bool operator==(const Gt &o) const { return std::memcmp(&v, &o.v, sizeof v) == 0; }
    bool operator!=(const Gt &o) const { return !(*this == o); }
};
static_assert(sizeof(Fr) == 32 && sizeof(G1) == 96 && sizeof(G2) == 192 && sizeof(Gt) == 384, "layouts must equal the crate's #[repr(C)] types");

// lib.rs:181-183
inline Gt pairing(const G1 &p, const G2 &q) { Gt r; check(bn254_pairing_batch(nullptr, &p.v, &q.v, &r.v, 1)); return r; }
// out[i] = pairing(p[i], q[i])
inline std::vector<Gt> pairing_batch(const std::vector<G1> &p, const std::vector<G2> &q) {
    if (p.size() != q.size()) throw std::invalid_argument("pairing_batch: length mismatch");
    std::vector<Gt> out(p.size());
    check(bn254_pairing_batch(nullptr, reinterpret_cast<const bn_g1 *>(p.data()), reinterpret_cast<const bn_g2 *>(q.data()),
                              reinterpret_cast<bn_gt *>(out.data()), p.size()));
    return out;
}
// fold(Gt::one(), acc * pairing(p, q))   (shootout/main.rs:11-16)
inline Gt pairing_product(const std::vector<G1> &p, const std::vector<G2> &q) {
    if (p.size() != q.size()) throw std::invalid_argument("pairing_product: length mismatch");
    Gt r;
    check(bn254_pairing_product(nullptr, reinterpret_cast<const bn_g1 *>(p.data()), reinterpret_cast<const bn_g2 *>(q.data()), p.size(), &r.v));
    return r;
}

// G2 points prepared ONCE for many pairings: the device-resident counterpart of the crate's internal G2Precomp (groups/mod.rs:472-483; `precompute`
// :557-588 runs in the constructor, on the GPU).  One point: shared by every p; several: point i is paired with p[i].
class PreparedG2 {
    bn254_g2_prepared *h_ = nullptr;
public:
    explicit PreparedG2(const std::vector<G2> &q) { check(bn254_g2_prepare(nullptr, reinterpret_cast<const bn_g2 *>(q.data()), q.size(), &h_)); }
    explicit PreparedG2(const G2 &q) { check(bn254_g2_prepare(nullptr, &q.v, 1, &h_)); }
    ~PreparedG2() { bn254_g2_prepared_destroy(h_); }
    PreparedG2(const PreparedG2 &) = delete;
    PreparedG2 &operator=(const PreparedG2 &) = delete;
    size_t size() const { return bn254_g2_prepared_count(h_); }
    size_t device_bytes() const { return bn254_g2_prepared_bytes(h_); }
    // out[i] == bn::pairing(p[i], q) (one prepared point) resp. bn::pairing(p[i], q[i])   (groups/mod.rs:486-519,764-771)
    std::vector<Gt> pairing_batch(const std::vector<G1> &p) const {
        std::vector<Gt> out(p.size());
        check(bn254_pairing_prepared_native_batch(nullptr, reinterpret_cast<const bn_g1 *>(p.data()), h_, reinterpret_cast<bn_gt *>(out.data()), p.size()));
        return out;
    }
    Gt pairing(const G1 &p) const { Gt r; check(bn254_pairing_prepared_native_batch(nullptr, &p.v, h_, &r.v, 1)); return r; }
    // == fold(Gt::one(), acc * bn::pairing(p[i], q[i]))   (shootout/main.rs:11-16): ONE final exponentiation, shared Miller accumulators
    Gt pairing_product(const std::vector<G1> &p) const {
        Gt r;
        check(bn254_pairing_product_prepared_native(nullptr, reinterpret_cast<const bn_g1 *>(p.data()), h_, p.size(), &r.v));
        return r;
    }
};

// tunables of the default context (BN254_OPT_* of bn254_hip.h; value < 0 restores the default derived from the device)
inline void set_option(int key, long value) { check(bn254_ctx_set_option(nullptr, key, value)); }
inline long get_option(int key) { long v = 0; check(bn254_ctx_get_option(nullptr, key, &v)); return v; }

// several GPUs of one node behind one handle (bn254_multi_*): shards of independent pairings, and the multi-pairing product with
// its single 384-byte-per-GPU exchange (RCCL all-gather over xGMI) and ONE final exponentiation
class MultiGpu {
    bn254_multi *m_ = nullptr;
public:
    // exchange: BN254_EXCHANGE_AUTO (RCCL when every rank has its own GPU), _PEER, _RCCL (throws instead of falling back)
    explicit MultiGpu(const std::vector<int> &devices, int exchange = BN254_EXCHANGE_AUTO) { check(bn254_multi_create_ex(devices.data(), (int)devices.size(), exchange, &m_)); }
    void set_option(int key, long value) { check(bn254_multi_set_option(m_, key, value)); }           // BN254_OPT_*, every rank's context
    int rank_numa_node(int rank) const { return bn254_multi_rank_numa_node(m_, rank); }
    ~MultiGpu() { bn254_multi_destroy(m_); }
    MultiGpu(const MultiGpu &) = delete;
    MultiGpu &operator=(const MultiGpu &) = delete;
    bool uses_rccl() const { return bn254_multi_exchange_kind(m_) == BN254_EXCHANGE_RCCL; }
    std::vector<Gt> pairing_batch(const std::vector<G1> &p, const std::vector<G2> &q) {
        if (p.size() != q.size()) throw std::invalid_argument("pairing_batch: length mismatch");
        std::vector<Gt> out(p.size());
        check(bn254_pairing_batch_multi(m_, reinterpret_cast<const bn_g1 *>(p.data()), reinterpret_cast<const bn_g2 *>(q.data()),
                                        reinterpret_cast<bn_gt *>(out.data()), p.size()));
        return out;
    }
    Gt pairing_product(const std::vector<G1> &p, const std::vector<G2> &q) {
        if (p.size() != q.size()) throw std::invalid_argument("pairing_product: length mismatch");
        Gt r;
        check(bn254_pairing_product_multi(m_, reinterpret_cast<const bn_g1 *>(p.data()), reinterpret_cast<const bn_g2 *>(q.data()), p.size(), &r.v));
        return r;
    }
};

}  // namespace bn

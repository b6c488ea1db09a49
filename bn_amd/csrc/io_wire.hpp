// Wire format of the reference crate on the GPU: big-endian canonical integers, Fq2 packed as the
// 512-bit integer c1*q + c0, points as tag byte (0 infinity / 4 affine) + coordinates; decoding validates exactly what the
// reference validates, in its order.  Reference: src/arith.rs:100-159 (U256/U512 codecs), :21-44 (U512::from), :65-88 (divrem),
// src/fields/fp.rs:24-36, src/fields/fq2.rs:31-53, src/groups/mod.rs:143-205.
// Batch records are fixed size (G1 65 bytes, G2 129 bytes); infinity is tag 0 followed by zero padding.
// The reference's own serialization test file is absent from the mount: parity is pinned by the code-derived oracle only.
#pragma once
#include "curve.hpp"

namespace bn254 {

enum { WIRE_OK = 0, WIRE_E_NOT_LESS_THAN_MODULUS = 1, WIRE_E_NOT_LESS_THAN_MODULUS_SQUARED = 2, WIRE_E_INVALID_LEADING_BYTE = 3,
       WIRE_E_NOT_ON_CURVE = 4, WIRE_E_NOT_IN_SUBGROUP = 5 };

// n 32-bit words (little-endian word order) <-> 4n big-endian bytes
template <int NW> BN_FN void words_to_be(const uint32_t *w, uint8_t *out) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t v = w[NW - 1 - i];
        out[4 * i] = (uint8_t)(v >> 24); out[4 * i + 1] = (uint8_t)(v >> 16); out[4 * i + 2] = (uint8_t)(v >> 8); out[4 * i + 3] = (uint8_t)v;
    }
}
template <int NW> BN_FN void be_to_words(const uint8_t *in, uint32_t *w) {
#pragma unroll
    for (int i = 0; i < NW; ++i)
        w[NW - 1 - i] = ((uint32_t)in[4 * i] << 24) | ((uint32_t)in[4 * i + 1] << 16) | ((uint32_t)in[4 * i + 2] << 8) | (uint32_t)in[4 * i + 3];
}
// internal (lazy, Montgomery 2^261) -> canonical integer words (fp.rs:15-22)
BN_FN void fe_to_raw_words(const Fe &a, uint32_t *w) { fe_pack_u32x8(fe_canonical(fe_mul(a, fe_const(k::RAW_ONE))), w); }
// any 256-bit integer -> internal form (value taken mod q)
BN_FN Fe fe_from_raw_words(const uint32_t *w) {
    Fe u = fe_unpack_u32x8(w);
    BN_IFB(u.vb = 6;)                       // a raw 256-bit integer is below 2^256 < 6q
    return fe_mul(u, fe_const(k::C_RAW_IN));
}
BN_FN bool words_lt_q(const uint32_t *w) {
    int64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { int64_t s = (int64_t)w[i] - (int64_t)k::Q32[i] + br; br = s >> 32; }
    return br != 0;
}
// r[0..16) = a[0..8) * q + c[0..8)      (U512::from, arith.rs:21-44)
BN_FN void mul_q_add(const uint32_t *a, const uint32_t *c, uint32_t *r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = i < 8 ? c[i] : 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { uint64_t t = (uint64_t)a[i] * k::Q32[j] + r[i + j] + carry; r[i + j] = (uint32_t)t; carry = t >> 32; }
#pragma unroll
        for (int j = i + 8; j < 16; ++j) { uint64_t t = (uint64_t)r[j] + carry; r[j] = (uint32_t)t; carry = t >> 32; }
    }
}
// 512-bit x = c1*q + c0 with c0, c1 < q ?  (the reference's bit-serial divrem, arith.rs:65-88, as an exact division:
// c0 = x mod q through the field arithmetic, c1 = (x - c0) * q^-1 mod 2^256, then verified by multiplying back)
BN_FN bool split_c1q_c0(const uint32_t *x, uint32_t *c1, uint32_t *c0) {
    Fe hi = fe_unpack_u32x8(x + 8), lo = fe_unpack_u32x8(x);
    BN_IFB(hi.vb = 6; lo.vb = 6;)
    Fe sum = fe_add(fe_mul(hi, fe_const(k::C_RAW_HI)), fe_mul(lo, fe_const(k::C_RAW_IN)));      // x mod q, internal form
    fe_to_raw_words(sum, c0);
    uint32_t d[8];
    int64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { int64_t s = (int64_t)x[i] - (int64_t)c0[i] + br; d[i] = (uint32_t)s; br = s >> 32; }
#pragma unroll
    for (int i = 0; i < 8; ++i) c1[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                                        // low 256 bits of d * q^-1
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j + i < 8; ++j) { uint64_t t = (uint64_t)d[i] * k::QINV256[j] + c1[i + j] + carry; c1[i + j] = (uint32_t)t; carry = t >> 32; }
    }
    uint32_t back[16];
    mul_q_add(c1, c0, back);
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) diff |= back[i] ^ x[i];
    return diff == 0 && words_lt_q(c1);
}

// ---- G1: 65 bytes -------------------------------------------------------------------------------------------------------
BN_FN void g1_encode_record(const uint32_t *p, uint8_t *out) {            // groups/mod.rs:143-164
    bool inf = words_all_zero(p + 16, 8);
    G1Aff<Fe> a = g1_to_affine(fe_from_u32x8(p), fe_from_u32x8(p + 8), fe_from_u32x8(p + 16));
    uint32_t w[8];
    uint8_t buf[65];
    buf[0] = inf ? 0 : 4;
    fe_to_raw_words(a.x, w); words_to_be<8>(w, buf + 1);
    fe_to_raw_words(a.y, w); words_to_be<8>(w, buf + 33);
    for (int i = 0; i < 65; ++i) out[i] = (inf && i > 0) ? 0 : buf[i];
}
BN_FN int g1_decode_record(const uint8_t *in, uint32_t *out) {            // groups/mod.rs:166-205
    uint32_t xw[8], yw[8];
    be_to_words<8>(in + 1, xw); be_to_words<8>(in + 33, yw);
    uint8_t tag = in[0];
    int status = WIRE_OK;
    Fe x = fe_from_raw_words(xw), y = fe_from_raw_words(yw);
    Fe rhs = fe_lc3<1, 1, 0>(fe_mul(fe_sqr(x), x), fe_const(k::G1_B), x);
    bool on_curve = fe_is_zero(fe_lc3<1, -1, 0>(fe_sqr(y), rhs, y));
    if (!on_curve) status = WIRE_E_NOT_ON_CURVE;
    if (!words_lt_q(yw)) status = WIRE_E_NOT_LESS_THAN_MODULUS;
    if (!words_lt_q(xw)) status = WIRE_E_NOT_LESS_THAN_MODULUS;
    if (tag != 0 && tag != 4) status = WIRE_E_INVALID_LEADING_BYTE;
    if (tag == 0) status = WIRE_OK;
    bool inf = tag == 0 || status != WIRE_OK;                             // failed records come back as G::zero()
    uint32_t o[24];
    fe_to_u32x8(fe_select(inf, x, fe_zero()), o);
    fe_to_u32x8(fe_select(inf, y, fe_one()), o + 8);
    fe_to_u32x8(fe_select(inf, fe_one(), fe_zero()), o + 16);
    for (int i = 0; i < 24; ++i) out[i] = o[i];
    return status;
}

// ---- G2: 129 bytes ------------------------------------------------------------------------------------------------------
BN_FN void fq2a_encode(const Fq2A &a, uint8_t *out) {                    // fq2.rs:31-38
    uint32_t c0[8], c1[8], v[16];
    fe_to_raw_words(a.c0, c0); fe_to_raw_words(a.c1, c1);
    mul_q_add(c1, c0, v);
    words_to_be<16>(v, out);
}
BN_FN bool fq2a_decode(const uint8_t *in, Fq2A &a) {                     // fq2.rs:40-53
    uint32_t v[16], c0[8], c1[8];
    be_to_words<16>(in, v);
    bool ok = split_c1q_c0(v, c1, c0);
    a = {fe_from_raw_words(c0), fe_from_raw_words(c1)};
    return ok;
}
BN_FN void g2_encode_record(const uint32_t *p, uint8_t *out) {
    bool inf = words_all_zero(p + 32, 16);
    G2Aff<Fq2A> a = g2_to_affine(f2_load((const Fq2A *)nullptr, p), f2_load((const Fq2A *)nullptr, p + 16), f2_load((const Fq2A *)nullptr, p + 32));
    uint8_t buf[129];
    buf[0] = inf ? 0 : 4;
    fq2a_encode(a.x, buf + 1); fq2a_encode(a.y, buf + 65);
    for (int i = 0; i < 129; ++i) out[i] = (inf && i > 0) ? 0 : buf[i];
}
BN_FN int g2_decode_record(const uint8_t *in, uint32_t *out) {
    uint8_t tag = in[0];
    Fq2A x, y;
    bool okx = fq2a_decode(in + 1, x), oky = fq2a_decode(in + 65, y);
    int status = WIRE_OK;
    using F = Fq2Field<Fq2A>;
    // p * (-1) + p == 0   (groups/mod.rs:183-194): the scalar r - 1 is the same for every lane, so its bits drive uniform branches
    Jac<F> p = {x, y, F::one()};
    Jac<F> acc = p;
#pragma unroll 1
    for (int i = 252; i >= 0; --i) {                                      // r - 1 has 254 bits; the top bit is consumed by acc = p
        acc = jac_double(acc);
        if ((k::R_MINUS_1[i >> 6] >> (i & 63)) & 1) acc = jac_add_flags(acc, p, false, false);
    }
    Jac<F> sum = jac_add_flags(acc, p, false, false);
    if (!F::is_zero(sum.z)) status = WIRE_E_NOT_IN_SUBGROUP;
    Fq2A rhs = f2_lc3<1, 1, 0>(f2_mul(f2_sqr(x), x), f2_const((const Fq2A *)nullptr, k::G2_B), x);
    if (!f2_is_zero(f2_lc3<1, -1, 0>(f2_sqr(y), rhs, y))) status = WIRE_E_NOT_ON_CURVE;
    if (!oky) status = WIRE_E_NOT_LESS_THAN_MODULUS_SQUARED;
    if (!okx) status = WIRE_E_NOT_LESS_THAN_MODULUS_SQUARED;
    if (tag != 0 && tag != 4) status = WIRE_E_INVALID_LEADING_BYTE;
    if (tag == 0) status = WIRE_OK;
    bool inf = tag == 0 || status != WIRE_OK;
    uint32_t o[48];
    f2_store(f2_select(inf, x, F::zero()), o);
    f2_store(f2_select(inf, y, F::one()), o + 16);
    f2_store(f2_select(inf, F::one(), F::zero()), o + 32);
    for (int i = 0; i < 48; ++i) out[i] = o[i];
    return status;
}

// ---- Fr records (fields/fp.rs:24-36 for Fr): 32 bytes big-endian of the canonical integer; decode rejects values >= r -----
// a * b / 2^256 mod r on 8 x u32 words (word-serial Montgomery; used only here and for the scalar extraction in curve.hpp)
BN_FN void fr_montmul(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t x = (uint64_t)a[i] * b[j] + t[j] + c;
            t[j] = (uint32_t)x; c = x >> 32;
        }
        uint64_t x = (uint64_t)t[8] + c;
        t[8] = (uint32_t)x; t[9] = (uint32_t)(x >> 32);
        uint32_t m = t[0] * k::FR_INV32;
        c = ((uint64_t)m * k::FR_MOD32[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            uint64_t y = (uint64_t)m * k::FR_MOD32[j] + t[j] + c;
            t[j - 1] = (uint32_t)y; c = y >> 32;
        }
        uint64_t y = (uint64_t)t[8] + c;
        t[7] = (uint32_t)y;
        t[8] = t[9] + (uint32_t)(y >> 32);
        t[9] = 0;
    }
    uint32_t d[8];
    int64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int64_t v = (int64_t)t[i] - (int64_t)k::FR_MOD32[i] + br;
        d[i] = (uint32_t)v; br = v >> 32;
    }
    bool ge = (t[8] != 0) || (br == 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = ge ? d[i] : t[i];
}
BN_FN bool words_lt_r(const uint32_t *w) {
    int64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { int64_t v = (int64_t)w[i] - (int64_t)k::FR_MOD32[i] + br; br = v >> 32; }
    return br != 0;
}
BN_FN void fr_encode_record(const uint32_t *km, uint8_t *out) {
    uint32_t raw[8];
    fr_from_mont(km, raw);
    words_to_be<8>(raw, out);
}
BN_FN int fr_decode_record(const uint8_t *in, uint32_t *out) {
    uint32_t w[8], r2[8];
    be_to_words<8>(in, w);
#pragma unroll
    for (int i = 0; i < 8; ++i) r2[i] = k::FR_R2_32[i];
    bool ok = words_lt_r(w);
    uint32_t m[8];
    fr_montmul(w, r2, m);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = ok ? m[i] : 0u;                  // rejected records come back as Fr::zero()
    return ok ? WIRE_OK : WIRE_E_NOT_LESS_THAN_MODULUS;
}

}  // namespace bn254

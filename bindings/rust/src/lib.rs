//! UNVERIFIED SOURCE (no rustc/cargo in the build image; never compiled).
//!
//! Batch GPU entry points for the `bn` crate (zcash-hackworks/bn v0.4.3) over the C ABI of `include/bn254_hip.h`.
//! `bn::{Fr, G1, G2, Gt}` are `#[repr(C)]` newtype chains down to `[u64; 4]` (src/lib.rs:15-17, 79-81, 122-124, 165-167), so a
//! slice of them is bit-for-bit an array of `bn_fr` / `bn_g1` / `bn_g2` / `bn_gt`: no conversion, no copies beyond the DMA.
//! Results are `==`-equal (and, for `Gt`, byte-equal) to what the crate computes on the CPU.
extern crate bn;

use bn::{Fr, Group, G1, G2, Gt};          // Group: `zero()` / `one()` of G1 and G2 are trait methods (src/lib.rs:56-77)
use std::os::raw::{c_int, c_long, c_void};

extern "C" {
    fn bn254_pairing_batch(ctx: *mut c_void, p: *const G1, q: *const G2, out: *mut Gt, n: usize) -> c_int;
    fn bn254_pairing_product(ctx: *mut c_void, p: *const G1, q: *const G2, n: usize, out: *mut Gt) -> c_int;
    fn bn254_g1_mul_batch(ctx: *mut c_void, p: *const G1, k: *const Fr, out: *mut G1, n: usize) -> c_int;
    fn bn254_g2_mul_batch(ctx: *mut c_void, p: *const G2, k: *const Fr, out: *mut G2, n: usize) -> c_int;
    fn bn254_gt_mul_batch(ctx: *mut c_void, a: *const Gt, b: *const Gt, out: *mut Gt, n: usize) -> c_int;
    fn bn254_gt_pow_batch(ctx: *mut c_void, a: *const Gt, k: *const Fr, out: *mut Gt, n: usize) -> c_int;
    fn bn254_g1_add_batch(ctx: *mut c_void, a: *const G1, b: *const G1, out: *mut G1, n: usize, negate_b: c_int) -> c_int;
    fn bn254_g2_add_batch(ctx: *mut c_void, a: *const G2, b: *const G2, out: *mut G2, n: usize, negate_b: c_int) -> c_int;
    fn bn254_g1_decode_batch(ctx: *mut c_void, bytes: *const u8, out: *mut G1, status: *mut i32, n: usize) -> c_int;
    fn bn254_g2_decode_batch(ctx: *mut c_void, bytes: *const u8, out: *mut G2, status: *mut i32, n: usize) -> c_int;
    fn bn254_gt_inverse_batch(ctx: *mut c_void, a: *const Gt, out: *mut Gt, n: usize) -> c_int;
    // prepared-G2 mode (the crate's internal G2Precomp, src/groups/mod.rs:472-483): 102 line coefficients per Q, then many P against them
    fn bn254_g2_precompute(ctx: *mut c_void, q: *const G2, coeffs: *mut EllCoeffs, n: usize) -> c_int;
    fn bn254_pairing_prepared_batch(ctx: *mut c_void, p: *const G1, coeffs: *const EllCoeffs, shared: c_int, out: *mut Gt, n: usize) -> c_int;
    // native prepared-G2 mode: the device-resident counterpart of G2Precomp behind an opaque handle (include/bn254_hip.h bn254_g2_prepare)
    fn bn254_g2_prepare(ctx: *mut c_void, q: *const G2, nq: usize, out: *mut *mut c_void) -> c_int;
    fn bn254_g2_prepared_destroy(prep: *mut c_void);
    fn bn254_g2_prepared_count(prep: *const c_void) -> usize;
    fn bn254_g2_prepared_bytes(prep: *const c_void) -> usize;
    fn bn254_pairing_prepared_native_batch(ctx: *mut c_void, p: *const G1, prep: *const c_void, out: *mut Gt, n: usize) -> c_int;
    fn bn254_pairing_product_prepared_native(ctx: *mut c_void, p: *const G1, prep: *const c_void, n: usize, out: *mut Gt) -> c_int;
    // wire format of the crate's Encodable / Decodable impls (src/groups/mod.rs:143-205, src/fields/fp.rs:24-36), fixed-size records and the stream
    fn bn254_fr_encode_batch(ctx: *mut c_void, k: *const Fr, out: *mut u8, n: usize) -> c_int;
    fn bn254_fr_decode_batch(ctx: *mut c_void, bytes: *const u8, out: *mut Fr, status: *mut i32, n: usize) -> c_int;
    fn bn254_g1_encode_batch(ctx: *mut c_void, p: *const G1, out: *mut u8, n: usize) -> c_int;
    fn bn254_g2_encode_batch(ctx: *mut c_void, p: *const G2, out: *mut u8, n: usize) -> c_int;
    fn bn254_g1_encode_stream(ctx: *mut c_void, p: *const G1, n: usize, out: *mut u8, cap: usize, written: *mut usize) -> c_int;
    fn bn254_g2_encode_stream(ctx: *mut c_void, p: *const G2, n: usize, out: *mut u8, cap: usize, written: *mut usize) -> c_int;
    fn bn254_g1_decode_stream(ctx: *mut c_void, bytes: *const u8, len: usize, out: *mut G1, status: *mut i32, max_points: usize, count: *mut usize, consumed: *mut usize) -> c_int;
    fn bn254_g2_decode_stream(ctx: *mut c_void, bytes: *const u8, len: usize, out: *mut G2, status: *mut i32, max_points: usize, count: *mut usize, consumed: *mut usize) -> c_int;
    // one node, several GPUs (include/bn254_hip.h "bn254_multi"): one context + host thread per device inside the library
    fn bn254_multi_create(devices: *const c_int, ndev: c_int, out: *mut *mut c_void) -> c_int;
    fn bn254_multi_create_ex(devices: *const c_int, ndev: c_int, exchange: c_int, out: *mut *mut c_void) -> c_int;     // -1 auto, 0 peer copies, 1 RCCL
    fn bn254_multi_set_option(m: *mut c_void, key: c_int, value: c_long) -> c_int;
    fn bn254_multi_rank_numa_node(m: *const c_void, rank: c_int) -> c_int;
    // tunables of a context (NULL = the default context of the current device): BN254_OPT_* of include/bn254_hip.h; value < 0 = default
    fn bn254_ctx_set_option(ctx: *mut c_void, key: c_int, value: c_long) -> c_int;
    fn bn254_ctx_get_option(ctx: *mut c_void, key: c_int, value: *mut c_long) -> c_int;
    fn bn254_ctx_get_option_raw(ctx: *mut c_void, key: c_int, value: *mut c_long) -> c_int;
    fn bn254_multi_destroy(m: *mut c_void);
    fn bn254_g2_prepare_multi(m: *mut c_void, q: *const G2, nq: usize, out: *mut *mut c_void) -> c_int;
    fn bn254_multi_prepared_destroy(prep: *mut c_void);
    fn bn254_multi_prepared_count(prep: *const c_void) -> usize;
    fn bn254_pairing_prepared_native_batch_multi(m: *mut c_void, p: *const G1, prep: *const c_void, out: *mut Gt, n: usize) -> c_int;
    fn bn254_pairing_product_prepared_native_multi(m: *mut c_void, p: *const G1, prep: *const c_void, n: usize, out: *mut Gt) -> c_int;
    fn bn254_pairing_batch_multi(m: *mut c_void, p: *const G1, q: *const G2, out: *mut Gt, n: usize) -> c_int;
    fn bn254_pairing_product_multi(m: *mut c_void, p: *const G1, q: *const G2, n: usize, out: *mut Gt) -> c_int;
}

/// One line-function coefficient of a prepared G2 point: the crate's `EllCoeffs { ell_0, ell_vw, ell_vv: Fq2 }` (src/groups/mod.rs:472-476) as the
/// C ABI lays it out (`bn_ell_coeffs`: three Fq2 = 3 x 8 u64 Montgomery limbs); the crate keeps the type private, so the binding carries its own.
#[derive(Clone, Copy)]
#[repr(C)]
pub struct EllCoeffs { pub ell_0: [u64; 8], pub ell_vw: [u64; 8], pub ell_vv: [u64; 8] }
/// `BN254_PREPARED_COEFFS` of include/bn254_hip.h: coefficients per prepared point (G2Precomp.coeffs, src/groups/mod.rs:478-483)
pub const PREPARED_COEFFS: usize = 102;
/// `BN254_PREPARED_NATIVE_LINES` / `BN254_PREPARED_NATIVE_BYTES`: lines and device bytes per point of a `PreparedG2`
pub const PREPARED_NATIVE_LINES: usize = 88;
pub const PREPARED_NATIVE_BYTES: usize = 33792;
/// `BN254_FR_WIRE_BYTES` / `BN254_G1_WIRE_BYTES` / `BN254_G2_WIRE_BYTES`
pub const FR_WIRE_BYTES: usize = 32;
pub const G1_WIRE_BYTES: usize = 65;
pub const G2_WIRE_BYTES: usize = 129;

// Thread safety: every host-buffer entry point of the C ABI may be called from any number of threads on one context (including the
// process-wide default context behind a NULL ctx), like the crate's own `pairing` (its types are `Send + Sync`,
// src/lib.rs:55-61): the batch entry points lease one of two pipeline slots per call (two callers overlap on the GPU, more queue),
// the others lock the context for the call.

/// `BN254_OPT_*` of include/bn254_hip.h: per-context policies whose defaults derive from the device's CU count
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
#[repr(i32)]
pub enum GpuOption {
    WavePairingMax = 1, WaveFeMax = 2, QuadMax = 3, MillerShared = 4, GtPowMode = 5, ProductChunk = 6, ProductPerWave = 7,
    ProductBfly = 8, RoundPairs = 9, PipelineChunk = 10, PipelineSlots = 11, StreamStopAtError = 12,
}
/// sets an option of the process-wide default context of the current HIP device; `None` restores the default
pub fn set_option(key: GpuOption, value: Option<i64>) -> Result<(), GpuError> {
    check(unsafe { bn254_ctx_set_option(std::ptr::null_mut(), key as c_int, value.unwrap_or(-1) as c_long) })
}
/// the explicitly set value of an option of the default context, `None` while its default is in effect
pub fn get_option_raw(key: GpuOption) -> Result<Option<i64>, GpuError> {
    let mut v: c_long = 0;
    check(unsafe { bn254_ctx_get_option_raw(std::ptr::null_mut(), key as c_int, &mut v) })?;
    Ok(if v < 0 { None } else { Some(v as i64) })
}
/// the effective value of an option of the default context
pub fn get_option(key: GpuOption) -> Result<i64, GpuError> {
    let mut v: c_long = 0;
    check(unsafe { bn254_ctx_get_option(std::ptr::null_mut(), key as c_int, &mut v) })?;
    Ok(v as i64)
}

/// Error code of the HIP engine: negative `BN254_E_*`, positive `hipError_t`.  There is no CPU fallback.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct GpuError(pub i32);

fn check(rc: c_int) -> Result<(), GpuError> { if rc == 0 { Ok(()) } else { Err(GpuError(rc)) } }

/// `out[i] = bn::pairing(p[i], q[i])` (src/lib.rs:181-183)
pub fn pairing_batch(p: &[G1], q: &[G2]) -> Result<Vec<Gt>, GpuError> {
    assert_eq!(p.len(), q.len());
    let mut out = vec![Gt::one(); p.len()];
    check(unsafe { bn254_pairing_batch(std::ptr::null_mut(), p.as_ptr(), q.as_ptr(), out.as_mut_ptr(), p.len()) })?;
    Ok(out)
}

/// `fold(Gt::one(), |acc, (p, q)| acc * pairing(p, q))` (shootout/main.rs:11-16) with a single final exponentiation
pub fn pairing_product(p: &[G1], q: &[G2]) -> Result<Gt, GpuError> {
    assert_eq!(p.len(), q.len());
    let mut out = Gt::one();
    check(unsafe { bn254_pairing_product(std::ptr::null_mut(), p.as_ptr(), q.as_ptr(), p.len(), &mut out) })?;
    Ok(out)
}

/// `out[i] = p[i] * k[i]`, returned normalized (src/lib.rs:88-95): equal to the crate's result under its projective `==`
pub fn g1_mul_batch(p: &[G1], k: &[Fr]) -> Result<Vec<G1>, GpuError> {
    assert_eq!(p.len(), k.len());
    let mut out = p.to_vec();
    check(unsafe { bn254_g1_mul_batch(std::ptr::null_mut(), p.as_ptr(), k.as_ptr(), out.as_mut_ptr(), p.len()) })?;
    Ok(out)
}

pub fn g2_mul_batch(p: &[G2], k: &[Fr]) -> Result<Vec<G2>, GpuError> {
    assert_eq!(p.len(), k.len());
    let mut out = p.to_vec();
    check(unsafe { bn254_g2_mul_batch(std::ptr::null_mut(), p.as_ptr(), k.as_ptr(), out.as_mut_ptr(), p.len()) })?;
    Ok(out)
}

/// `out[i] = a[i] * b[i]` (src/lib.rs:175-179)
pub fn gt_mul_batch(a: &[Gt], b: &[Gt]) -> Result<Vec<Gt>, GpuError> {
    assert_eq!(a.len(), b.len());
    let mut out = a.to_vec();
    check(unsafe { bn254_gt_mul_batch(std::ptr::null_mut(), a.as_ptr(), b.as_ptr(), out.as_mut_ptr(), a.len()) })?;
    Ok(out)
}

/// `out[i] = a[i].pow(k[i])` (src/lib.rs:171)
pub fn gt_pow_batch(a: &[Gt], k: &[Fr]) -> Result<Vec<Gt>, GpuError> {
    assert_eq!(a.len(), k.len());
    let mut out = a.to_vec();
    check(unsafe { bn254_gt_pow_batch(std::ptr::null_mut(), a.as_ptr(), k.as_ptr(), out.as_mut_ptr(), a.len()) })?;
    Ok(out)
}

/// `out[i] = a[i] + b[i]` (`sub`: `a[i] - b[i]`), src/lib.rs:103-111 - the crate's own Jacobian limbs
pub fn g1_add_batch(a: &[G1], b: &[G1], sub: bool) -> Result<Vec<G1>, GpuError> {
    assert_eq!(a.len(), b.len());
    let mut out = a.to_vec();
    check(unsafe { bn254_g1_add_batch(std::ptr::null_mut(), a.as_ptr(), b.as_ptr(), out.as_mut_ptr(), a.len(), sub as c_int) })?;
    Ok(out)
}

pub fn g2_add_batch(a: &[G2], b: &[G2], sub: bool) -> Result<Vec<G2>, GpuError> {
    assert_eq!(a.len(), b.len());
    let mut out = a.to_vec();
    check(unsafe { bn254_g2_add_batch(std::ptr::null_mut(), a.as_ptr(), b.as_ptr(), out.as_mut_ptr(), a.len(), sub as c_int) })?;
    Ok(out)
}

/// Batch `Decodable` for G2 (src/groups/mod.rs:162-205): `records` holds 129-byte records (what bincode yields for a finite
/// point); `Err(code)` per record carries the crate's error in order of appearance: 1 "integer is not less than modulus",
/// 2 "integer not less than modulus squared", 3 "invalid leading byte", 4 "point is not on the curve", 5 "point is not in the subgroup"
pub fn g2_decode_batch(records: &[u8]) -> Result<Vec<Result<G2, i32>>, GpuError> {
    assert_eq!(records.len() % G2_WIRE_BYTES, 0);
    let n = records.len() / G2_WIRE_BYTES;
    let (mut out, mut status) = (vec![G2::zero(); n], vec![0i32; n]);
    check(unsafe { bn254_g2_decode_batch(std::ptr::null_mut(), records.as_ptr(), out.as_mut_ptr(), status.as_mut_ptr(), n) })?;
    Ok(out.into_iter().zip(status).map(|(p, s)| if s == 0 { Ok(p) } else { Err(s) }).collect())
}

pub fn g1_decode_batch(records: &[u8]) -> Result<Vec<Result<G1, i32>>, GpuError> {
    assert_eq!(records.len() % G1_WIRE_BYTES, 0);
    let n = records.len() / G1_WIRE_BYTES;
    let (mut out, mut status) = (vec![G1::zero(); n], vec![0i32; n]);
    check(unsafe { bn254_g1_decode_batch(std::ptr::null_mut(), records.as_ptr(), out.as_mut_ptr(), status.as_mut_ptr(), n) })?;
    Ok(out.into_iter().zip(status).map(|(p, s)| if s == 0 { Ok(p) } else { Err(s) }).collect())
}

/// `q.to_affine().precompute()` for every q (src/groups/mod.rs:557-588; q must not be infinity): 102 coefficients per point, in schedule order
pub fn g2_precompute(q: &[G2]) -> Result<Vec<EllCoeffs>, GpuError> {
    let mut out = vec![EllCoeffs { ell_0: [0; 8], ell_vw: [0; 8], ell_vv: [0; 8] }; q.len() * PREPARED_COEFFS];
    check(unsafe { bn254_g2_precompute(std::ptr::null_mut(), q.as_ptr(), out.as_mut_ptr(), q.len()) })?;
    Ok(out)
}

/// `final_exponentiation(prepared.miller_loop(p[i]))` (src/groups/mod.rs:486-519, 768) against ONE prepared point (`coeffs.len() == 102`) or one
/// per p (`coeffs.len() == 102 * p.len()`)
pub fn pairing_prepared_batch(p: &[G1], coeffs: &[EllCoeffs]) -> Result<Vec<Gt>, GpuError> {
    let shared = coeffs.len() == PREPARED_COEFFS;
    assert!(shared || coeffs.len() == PREPARED_COEFFS * p.len());
    let mut out = vec![Gt::one(); p.len()];
    check(unsafe { bn254_pairing_prepared_batch(std::ptr::null_mut(), p.as_ptr(), coeffs.as_ptr(), shared as c_int, out.as_mut_ptr(), p.len()) })?;
    Ok(out)
}

/// G2 points prepared ONCE for many pairings (a verification key): the device-native counterpart of the crate's internal `G2Precomp`
/// (src/groups/mod.rs:472-483, `precompute` :557-588) - the line functions of the Miller loop in the form the kernels consume, resident in GPU
/// memory (33 792 bytes per point) on the default context's device.  Immutable after creation, hence `Sync`.
pub struct PreparedG2(*mut c_void);
unsafe impl Send for PreparedG2 {}
unsafe impl Sync for PreparedG2 {}

impl PreparedG2 {
    /// `q.len() == 1`: one point for every `p`; otherwise point `i` is paired with `p[i]`
    pub fn new(q: &[G2]) -> Result<PreparedG2, GpuError> {
        let mut h = std::ptr::null_mut();
        check(unsafe { bn254_g2_prepare(std::ptr::null_mut(), q.as_ptr(), q.len(), &mut h) })?;
        Ok(PreparedG2(h))
    }
    pub fn len(&self) -> usize { unsafe { bn254_g2_prepared_count(self.0) } }
    pub fn device_bytes(&self) -> usize { unsafe { bn254_g2_prepared_bytes(self.0) } }
    /// `out[i] = bn::pairing(p[i], q)` for a one-point handle, `bn::pairing(p[i], q[i])` otherwise (src/lib.rs:181-183 through
    /// src/groups/mod.rs:486-519: `precompute` happened in `new`)
    pub fn pairing_batch(&self, p: &[G1]) -> Result<Vec<Gt>, GpuError> {
        assert!(self.len() == 1 || p.len() <= self.len());
        let mut out = vec![Gt::one(); p.len()];
        check(unsafe { bn254_pairing_prepared_native_batch(std::ptr::null_mut(), p.as_ptr(), self.0, out.as_mut_ptr(), p.len()) })?;
        Ok(out)
    }
    /// `fold(Gt::one(), |acc, i| acc * bn::pairing(p[i], q[i]))` (shootout/main.rs:11-16) with ONE final exponentiation
    pub fn pairing_product(&self, p: &[G1]) -> Result<Gt, GpuError> {
        assert!(self.len() == 1 || p.len() <= self.len());
        let mut out = Gt::one();
        check(unsafe { bn254_pairing_product_prepared_native(std::ptr::null_mut(), p.as_ptr(), self.0, p.len(), &mut out) })?;
        Ok(out)
    }
}
impl Drop for PreparedG2 {
    fn drop(&mut self) { unsafe { bn254_g2_prepared_destroy(self.0) } }
}

/// the crate's `Encodable for Fr` (src/fields/fp.rs:24-36): 32 big-endian bytes per scalar
pub fn fr_encode_batch(k: &[Fr]) -> Result<Vec<u8>, GpuError> {
    let mut out = vec![0u8; k.len() * FR_WIRE_BYTES];
    check(unsafe { bn254_fr_encode_batch(std::ptr::null_mut(), k.as_ptr(), out.as_mut_ptr(), k.len()) })?;
    Ok(out)
}

/// `Decodable for Fr`: `Err(1)` = "integer is not less than modulus"
pub fn fr_decode_batch(records: &[u8]) -> Result<Vec<Result<Fr, i32>>, GpuError> {
    assert_eq!(records.len() % FR_WIRE_BYTES, 0);
    let n = records.len() / FR_WIRE_BYTES;
    let (mut out, mut status) = (vec![Fr::zero(); n], vec![0i32; n]);
    check(unsafe { bn254_fr_decode_batch(std::ptr::null_mut(), records.as_ptr(), out.as_mut_ptr(), status.as_mut_ptr(), n) })?;
    Ok(out.into_iter().zip(status).map(|(k, s)| if s == 0 { Ok(k) } else { Err(s) }).collect())
}

/// fixed-size records `[4][x][y]` (infinity: tag 0 and padding), 65 bytes per G1 point
pub fn g1_encode_batch(p: &[G1]) -> Result<Vec<u8>, GpuError> {
    let mut out = vec![0u8; p.len() * G1_WIRE_BYTES];
    check(unsafe { bn254_g1_encode_batch(std::ptr::null_mut(), p.as_ptr(), out.as_mut_ptr(), p.len()) })?;
    Ok(out)
}

pub fn g2_encode_batch(p: &[G2]) -> Result<Vec<u8>, GpuError> {
    let mut out = vec![0u8; p.len() * G2_WIRE_BYTES];
    check(unsafe { bn254_g2_encode_batch(std::ptr::null_mut(), p.as_ptr(), out.as_mut_ptr(), p.len()) })?;
    Ok(out)
}

/// the crate's own variable-length stream (what `bincode::encode` yields for a sequence of points: infinity is the lone byte 0)
pub fn g1_encode_stream(p: &[G1]) -> Result<Vec<u8>, GpuError> {
    let mut out = vec![0u8; p.len() * G1_WIRE_BYTES];
    let mut written = 0usize;
    check(unsafe { bn254_g1_encode_stream(std::ptr::null_mut(), p.as_ptr(), p.len(), out.as_mut_ptr(), out.len(), &mut written) })?;
    out.truncate(written);
    Ok(out)
}

pub fn g2_encode_stream(p: &[G2]) -> Result<Vec<u8>, GpuError> {
    let mut out = vec![0u8; p.len() * G2_WIRE_BYTES];
    let mut written = 0usize;
    check(unsafe { bn254_g2_encode_stream(std::ptr::null_mut(), p.as_ptr(), p.len(), out.as_mut_ptr(), out.len(), &mut written) })?;
    out.truncate(written);
    Ok(out)
}

/// up to `max_points` points from the crate's stream format; returns the points (a rejected record comes back as `Err(status)`, the decoder
/// goes on with the next record unless `GpuOption::StreamStopAtError` is set) and the number of bytes consumed
pub fn g1_decode_stream(bytes: &[u8], max_points: usize) -> Result<(Vec<Result<G1, i32>>, usize), GpuError> {
    let (mut out, mut status) = (vec![G1::zero(); max_points], vec![0i32; max_points]);
    let (mut count, mut consumed) = (0usize, 0usize);
    check(unsafe { bn254_g1_decode_stream(std::ptr::null_mut(), bytes.as_ptr(), bytes.len(), out.as_mut_ptr(), status.as_mut_ptr(), max_points, &mut count, &mut consumed) })?;
    out.truncate(count); status.truncate(count);
    Ok((out.into_iter().zip(status).map(|(p, s)| if s == 0 { Ok(p) } else { Err(s) }).collect(), consumed))
}

pub fn g2_decode_stream(bytes: &[u8], max_points: usize) -> Result<(Vec<Result<G2, i32>>, usize), GpuError> {
    let (mut out, mut status) = (vec![G2::zero(); max_points], vec![0i32; max_points]);
    let (mut count, mut consumed) = (0usize, 0usize);
    check(unsafe { bn254_g2_decode_stream(std::ptr::null_mut(), bytes.as_ptr(), bytes.len(), out.as_mut_ptr(), status.as_mut_ptr(), max_points, &mut count, &mut consumed) })?;
    out.truncate(count); status.truncate(count);
    Ok((out.into_iter().zip(status).map(|(p, s)| if s == 0 { Ok(p) } else { Err(s) }).collect(), consumed))
}

#[cfg(test)]
mod tests {
    // mirrors shootout/main.rs: the GPU fold equals the CPU fold
    use super::*;
    #[test]
    fn product_matches_cpu_fold() {
        let (mut a, mut b) = (G1::one(), G2::one());
        let c = Fr::from_str("1901").unwrap().inverse().unwrap();
        let d = Fr::from_str("2344").unwrap().inverse().unwrap();
        let (mut ps, mut qs, mut acc) = (vec![], vec![], Gt::one());
        for _ in 0..64 {
            acc = acc * bn::pairing(a, b);
            ps.push(a); qs.push(b);
            a = a * c; b = b * d;
        }
        assert!(pairing_product(&ps, &qs).unwrap() == acc);
        assert!(pairing_batch(&ps, &qs).unwrap().into_iter().fold(Gt::one(), |x, y| x * y) == acc);
    }
}


/// `out[i] = a[i].inverse()` (src/lib.rs:172)
pub fn gt_inverse_batch(a: &[Gt]) -> Result<Vec<Gt>, GpuError> {
    let mut out = a.to_vec();
    check(unsafe { bn254_gt_inverse_batch(std::ptr::null_mut(), a.as_ptr(), out.as_mut_ptr(), a.len()) })?;
    Ok(out)
}

/// All (or some) GPUs of one node behind one handle: independent pairings are sharded by contiguous ranges, the multi-pairing
/// product exchanges ONE 384-byte partial per GPU (RCCL all-gather over xGMI) and runs a single final exponentiation.
pub struct MultiGpu(*mut c_void);
unsafe impl Send for MultiGpu {}
unsafe impl Sync for MultiGpu {}          // the handle locks internally: one multi-device call at a time

impl MultiGpu {
    /// `devices`: HIP device index of every rank, e.g. `&[0, 1, 2, 3, 4, 5, 6, 7]`
    pub fn new(devices: &[i32]) -> Result<MultiGpu, GpuError> {
        let mut h = std::ptr::null_mut();
        check(unsafe { bn254_multi_create(devices.as_ptr(), devices.len() as c_int, &mut h) })?;
        Ok(MultiGpu(h))
    }
    /// the same with the exchange of the product forced: `Some(false)` peer copies, `Some(true)` RCCL (error instead of a fall-back)
    pub fn with_exchange(devices: &[i32], rccl: Option<bool>) -> Result<MultiGpu, GpuError> {
        let mut h = std::ptr::null_mut();
        let kind = match rccl { None => -1, Some(false) => 0, Some(true) => 1 };
        check(unsafe { bn254_multi_create_ex(devices.as_ptr(), devices.len() as c_int, kind, &mut h) })?;
        Ok(MultiGpu(h))
    }
    /// an option on every rank's context
    pub fn set_option(&self, key: GpuOption, value: Option<i64>) -> Result<(), GpuError> {
        check(unsafe { bn254_multi_set_option(self.0, key as c_int, value.unwrap_or(-1) as c_long) })
    }
    /// NUMA node the host thread of `rank` is pinned to during a call (None: not pinned)
    pub fn rank_numa_node(&self, rank: usize) -> Option<i32> {
        let n = unsafe { bn254_multi_rank_numa_node(self.0, rank as c_int) };
        if n < 0 { None } else { Some(n) }
    }
    /// `out[i] = bn::pairing(p[i], q[i])`, 2^20 pairings over 8 GPUs = BASELINE configs[2]
    pub fn pairing_batch(&self, p: &[G1], q: &[G2]) -> Result<Vec<Gt>, GpuError> {
        assert_eq!(p.len(), q.len());
        let mut out = vec![Gt::one(); p.len()];
        check(unsafe { bn254_pairing_batch_multi(self.0, p.as_ptr(), q.as_ptr(), out.as_mut_ptr(), p.len()) })?;
        Ok(out)
    }
    /// the fold of shootout/main.rs:11-16 over all pairs (BASELINE configs[3])
    pub fn pairing_product(&self, p: &[G1], q: &[G2]) -> Result<Gt, GpuError> {
        assert_eq!(p.len(), q.len());
        let mut out = Gt::one();
        check(unsafe { bn254_pairing_product_multi(self.0, p.as_ptr(), q.as_ptr(), p.len(), &mut out) })?;
        Ok(out)
    }
}
/// G2 points prepared on the GPUs of a `MultiGpu` (one point: on every GPU; several: sharded like the pairings they will meet)
pub struct MultiPreparedG2<'a> { h: *mut c_void, gpus: &'a MultiGpu }
impl MultiGpu {
    pub fn prepare_g2(&self, q: &[G2]) -> Result<MultiPreparedG2, GpuError> {
        let mut h = std::ptr::null_mut();
        check(unsafe { bn254_g2_prepare_multi(self.0, q.as_ptr(), q.len(), &mut h) })?;
        Ok(MultiPreparedG2 { h, gpus: self })
    }
}
impl<'a> MultiPreparedG2<'a> {
    pub fn len(&self) -> usize { unsafe { bn254_multi_prepared_count(self.h) } }
    /// `out[i] = bn::pairing(p[i], q)` (one prepared point) resp. `bn::pairing(p[i], q[i])` (`p.len() == self.len()`), sharded over the GPUs
    pub fn pairing_batch(&self, p: &[G1]) -> Result<Vec<Gt>, GpuError> {
        let mut out = vec![Gt::one(); p.len()];
        check(unsafe { bn254_pairing_prepared_native_batch_multi(self.gpus.0, p.as_ptr(), self.h, out.as_mut_ptr(), p.len()) })?;
        Ok(out)
    }
    /// the fold of shootout/main.rs:11-16 over the prepared points, sharded: one 384-byte exchange, ONE final exponentiation
    pub fn pairing_product(&self, p: &[G1]) -> Result<Gt, GpuError> {
        let mut out = Gt::one();
        check(unsafe { bn254_pairing_product_prepared_native_multi(self.gpus.0, p.as_ptr(), self.h, p.len(), &mut out) })?;
        Ok(out)
    }
}
impl<'a> Drop for MultiPreparedG2<'a> {
    fn drop(&mut self) { unsafe { bn254_multi_prepared_destroy(self.h) } }
}
impl Drop for MultiGpu {
    fn drop(&mut self) { unsafe { bn254_multi_destroy(self.0) } }
}

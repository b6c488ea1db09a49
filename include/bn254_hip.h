/*
 * bn254_hip.h - C ABI of the MI355X-native batched BN254 optimal-ate pairing engine.
 *
 * Drop-in boundary for the `pairing()` hot path of the reference crate zcash-hackworks/bn v0.4.3.  The reference has no
 * FFI of its own: its boundary is the public Rust API, whose types are all #[repr(C)] newtype chains, which fixes the
 * C layouts below (citations are file:line under the reference's src/):
 *
 *   bn_fr  = bn::Fr -> fields::Fr -> U256 -> [u64;4]        lib.rs:15-17, fields/fp.rs:11-13, arith.rs:9-11
 *   bn_g1  = bn::G1 -> G<G1Params>{x,y,z: Fq}               lib.rs:79-81, groups/mod.rs:36-41
 *   bn_g2  = bn::G2 -> G<G2Params>{x,y,z: Fq2{c0,c1}}       lib.rs:122-124, fields/fq2.rs:24-29
 *   bn_gt  = bn::Gt -> Fq12{c0,c1: Fq6{c0,c1,c2: Fq2}}      lib.rs:165-167, fields/fq12.rs:26-31, fields/fq6.rs:42-48
 *
 * Every Fq/Fr is 4 little-endian u64 limbs holding the Montgomery image a*2^256 mod m, always < m (canonical), exactly
 * the bytes the reference keeps in memory.  Points are Jacobian (X/Z^2, Y/Z^3); infinity is z == 0.
 *
 * Semantics replaced:
 *   bn254_pairing_batch    out[i] = bn::pairing(p[i], q[i])                         lib.rs:181-183, groups/mod.rs:764-771
 *   bn254_pairing_product  fold(Gt::one(), |acc,(p,q)| acc * pairing(p,q))          shootout/main.rs:11-16, lib.rs:175-179
 *   bn254_g1_mul_batch     out[i] = normalize(p[i] * k[i])                          lib.rs:116-120,88-95, groups/mod.rs:250-270
 *   bn254_g2_mul_batch     same over G2                                             lib.rs:159-163,131-138
 *   bn254_g1/g2_add_batch  out[i] = a[i] + b[i] / a[i] - b[i] (raw Jacobian limbs)       lib.rs:103-114,146-157, groups/mod.rs:275-347
 *   bn254_g2_precompute    coeffs[i][0..102) = q[i].to_affine().precompute().coeffs   groups/mod.rs:557-588 (Q != infinity)
 *   bn254_pairing_prepared_batch  out[i] = final_exponentiation(prepared.miller_loop(p[i]))   groups/mod.rs:486-519,768
 *   bn254_gt_mul_batch     out[i] = a[i] * b[i]                                     lib.rs:175-179, fields/fq12.rs:295-307
 *   bn254_gt_pow_batch     out[i] = a[i].pow(k[i])                                  lib.rs:171, fields/mod.rs:35-46
 * Outputs are bit-identical to the reference's on the same inputs (pairing values are canonical field elements; scalar
 * multiples are compared after `normalize()` because Jacobian coordinates depend on the addition chain).
 *
 * Error behaviour: the reference path is infallible for valid points (the `expect` at groups/mod.rs:768 cannot fire);
 * a point at infinity in either argument gives Gt::one() (groups/mod.rs:766).  So the only failures are device failures:
 * every function returns 0 on success or a negative BN254_E_* / positive hipError_t code; nothing panics, throws or
 * aborts across this boundary.  Inputs are trusted to be valid subgroup points exactly as the Rust type system guarantees
 * for G1/G2 values; behaviour on other limb patterns is unspecified (but memory safe).
 *
 * Sizes: n is limited by device memory only.  Batches are cut internally into sub-launches of at most one machine round, and the
 * context-owned tables (final exponentiation: 4 KB, Gt::pow: 14.8 KB per pairing) are sized for ONE round - 264 MB / 970 MB on an
 * MI355X whatever n is.  Up to 3584 pairings (3328 final exponentiations) per call (the tail of every multi-pairing: exactly one) run one per
 * WAVE instead of one per lane pair: a pairing in 1.0 ms instead of 4.2 ms, a final exponentiation in 0.5 ms instead of 2.0 ms; from
 * there up to 16384 per call a pairing runs on FOUR lanes (2.7 ms); above, on lane pairs (BN254_OPT_* below move the thresholds).
 *
 * Ownership: the caller owns every buffer passed in; the library owns device memory and streams inside a context and keeps
 * no pointer after a call returns.  A context is bound to one GPU.  There is NO CPU fallback: without a usable MI355X the
 * calls fail with BN254_E_NO_DEVICE.
 *
 * Threading (the reference's `pairing` is a pure function and its types are Send + Sync, lib.rs:55-61):
 *   - every HOST-BUFFER entry point is safe to call from any number of threads on the same context, including ctx == NULL
 *     (a process-wide default context per HIP device).  bn254_pairing_batch and bn254_g{1,2}_mul_batch arbitrate per pipeline
 *     slot: two callers with batches of up to one machine round (256 pairings per CU: 2^16 on an MI355X) run concurrently on two
 *     streams (the number of streams the GPU overlaps without loss), further callers and multi-chunk batches queue; every other
 *     entry point serialises its callers on the context.  Use one context per thread (or bn254_multi_*) for more overlap;
 *     bn254_ctx_set_option is atomic, but set options before concurrent use: a call in flight may run some
 *     of its launches under the old and some under the new setting (same bytes either way);
 *   - the *_dev entry points are asynchronous on the caller's stream.  Context-owned scratch (the final-exponentiation table,
 *     the product workspace) is ordered across streams with events, so calls on different streams of one context are safe
 *     and serialise on that scratch; the caller still owns the ordering of its OWN buffers between streams.
 */
#ifndef BN254_HIP_H
#define BN254_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } bn_fr;                    /* 32 B  */
typedef struct { uint64_t x[4], y[4], z[4]; } bn_g1;        /* 96 B  */
typedef struct { uint64_t x[8], y[8], z[8]; } bn_g2;        /* 192 B: each coordinate = (c0[4], c1[4]) */
typedef struct { uint64_t c[48]; } bn_gt;                   /* 384 B: c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1 */

/* one line-function coefficient of a prepared G2 point: the reference's EllCoeffs {ell_0, ell_vw, ell_vv: Fq2}
   (groups/mod.rs:472-476); a prepared point is 102 of them in schedule order (G2Precomp.coeffs, groups/mod.rs:478-483) */
typedef struct { uint64_t ell_0[8], ell_vw[8], ell_vv[8]; } bn_ell_coeffs;   /* 192 B */
#define BN254_PREPARED_COEFFS 102

typedef struct bn254_ctx bn254_ctx;

enum {
    BN254_OK = 0,
    BN254_E_NO_DEVICE = -1,     /* no HIP device / device index out of range */
    BN254_E_BAD_ARG = -2,       /* null pointer with n > 0, n too large */
    BN254_E_ALLOC = -3,         /* device allocation failed */
    BN254_E_COMM = -4,          /* RCCL / peer exchange of the multi-device product failed */
    BN254_E_INTERNAL = -5       /* an unexpected C++ exception was stopped at the boundary (nothing unwinds across it) */
    /* positive values are hipError_t codes */
};

/* ---- contexts ------------------------------------------------------------------------------------------------------ */
int bn254_device_count(void);
int bn254_ctx_create(int device, bn254_ctx **out);
void bn254_ctx_destroy(bn254_ctx *ctx);
const char *bn254_error_string(int code);
/* Kept for ABI compatibility: 1 = one pairing per lane PAIR (with its one-per-wave and four-lane siblings, chosen by batch size) is the only
   mapping; 0 (one pairing per lane: the test double of rounds 1-4, now tests/testdouble/) is rejected with BN254_E_BAD_ARG. */
int bn254_ctx_set_mapping(bn254_ctx *ctx, int mapping);

/* ---- tunables ---------------------------------------------------------------------------------------------------------
   Every policy of the host side is a per-context option whose default is derived from the device the context is bound to (its
   compute-unit count, `CUs` below); the call paths read nothing from the environment.  value < 0 restores the default;
   bn254_ctx_get_option reports the EFFECTIVE value - or -1 for the four options whose default is decided PER CALL from the batch
   size (BN254_OPT_PRODUCT_CHUNK / _PER_WAVE / _BFLY, BN254_OPT_PIPELINE_CHUNK) while none is set explicitly.  The size options
   (BN254_OPT_WAVE_PAIRING_MAX, _WAVE_FE_MAX, _QUAD_MAX, _ROUND_PAIRS, _PIPELINE_CHUNK) accept at most 2^22: what one launch addresses
   with 32-bit offsets; a value above is rejected with BN254_E_BAD_ARG (larger batches are cut into sub-launches regardless).  ctx == NULL addresses the default context of the current device.  An option
   may be changed at any time; a call in flight may see the old or the new value between two of its launches - harmless, because
   every option selects between kernels that return the same bytes (the one exception is stated at BN254_OPT_GT_POW_MODE).
   For experiments only, the variables BN254_WAVE_PAIRING_MAX, BN254_WAVE_FE_MAX, BN254_QUAD_MAX, BN254_MILLER_SHARED, BN254_GT_POW_MODE,
   BN254_PRODUCT_CHUNK / _PER_WAVE / _BFLY, BN254_ROUND_PAIRS, BN254_PIPELINE_CHUNK / _SLOTS, BN254_STREAM_STOP_AT_ERROR, BN254_MULTI_EXCHANGE (rccl | peer) and
   BN254_MULTI_AFFINITY (0: no thread pinning) are
   read ONCE per process, when the first context is created, and seed the options of every context created afterwards. */
enum {
    BN254_OPT_WAVE_PAIRING_MAX = 1, /* pairings (or Miller loops that only meet a final exponentiation) per call up to which ONE PER WAVE
                                       runs (csrc/bn254_kernels_w.hip).  Default 14 x CUs (3584): where it is level with the four-lane
                                       kernels on 256 CUs (profiles/r04_wave_latency.json), 13 workgroups of 11.5 KB LDS per CU */
    BN254_OPT_WAVE_FE_MAX = 2,      /* the same for final exponentiations.  Default 13 x CUs */
    BN254_OPT_QUAD_MAX = 3,         /* pairings per call up to which (and above the two options before) a pairing is spread over FOUR lanes
                                       instead of two (csrc/bn254_kernels_q.hip): the faster mapping while lane pairs would leave SIMDs
                                       empty.  Default 64 x CUs (16384): half a machine round of lane pairs */
    BN254_OPT_MILLER_SHARED = 4,    /* pairs per lane pair on ONE accumulator in the multi-pairing's Miller loop: 1, 2 or 4; 0 (default):
                                       4 from four machine rounds of pairs on, 2 from two, else 1 */
    BN254_OPT_GT_POW_MODE = 5,      /* Gt::pow chain: 0 (default) Frobenius decomposition - exact for elements whose ORDER DIVIDES r, which
                                       is everything the reference's Gt can hold; 2 one-dimensional cyclotomic chain - exact for ANY
                                       cyclotomic element; 1 the reference's general chain for everything.  THE ONE OPTION THAT CAN
                                       CHANGE RESULTS: only for inputs outside the r-torsion, which the reference's typed API cannot
                                       produce (see bn254_gt_pow_batch) */
    BN254_OPT_PRODUCT_CHUNK = 6,    /* shape of the one-launch Fq12 product tree: values per lane pair (1..4096), */
    BN254_OPT_PRODUCT_PER_WAVE = 7, /* live lane pairs per wave (1..32), */
    BN254_OPT_PRODUCT_BFLY = 8,     /* butterfly levels inside a wave (0..5).  Defaults: by size, profiles/r03p_product_shape_sweep.txt */
    BN254_OPT_ROUND_PAIRS = 9,      /* pairings per launch of the lane-pair kernels ("one machine round").  Default 256 x CUs: two waves
                                       on every SIMD; larger batches run as equal sub-launches of at most this size */
    BN254_OPT_PIPELINE_CHUNK = 10,  /* pairings per chunk of the pipelined host-buffer path.  Default: the sub-launch size */
    BN254_OPT_PIPELINE_SLOTS = 11,  /* chunks in flight, 1..4.  Default 2: the number of streams the GPU overlaps without loss */
    BN254_OPT_STREAM_STOP_AT_ERROR = 12, /* bn254_g{1,2}_decode_stream: 1 = the crate's own behaviour - its Decodable returns Err at the first bad
                                       record (groups/mod.rs:165-175) -: `count` ends WITH the first record whose status is non-zero and
                                       `consumed` behind it (out[] and status[] beyond `count` are unspecified: the batch decoder has
                                       already run over the records that follow); 0 (default): decode every record, report every status */
    BN254_OPT_COUNT_ = 13
};
int bn254_ctx_set_option(bn254_ctx *ctx, int key, long value);
int bn254_ctx_get_option(bn254_ctx *ctx, int key, long *value);
/* the RAW state of an option: the explicitly set value, or -1 while the default is in effect (what a scoped set / restore must save) */
int bn254_ctx_get_option_raw(bn254_ctx *ctx, int key, long *value);

/* ---- host-buffer entry points (what a binding of the reference's API calls) ------------------------------------------ */
/* ctx == NULL uses a process-wide default context on the current HIP device. */
int bn254_pairing_batch(bn254_ctx *ctx, const bn_g1 *p, const bn_g2 *q, bn_gt *out, size_t n);
int bn254_pairing_product(bn254_ctx *ctx, const bn_g1 *p, const bn_g2 *q, size_t n, bn_gt *out);
int bn254_g1_mul_batch(bn254_ctx *ctx, const bn_g1 *p, const bn_fr *k, bn_g1 *out, size_t n);
int bn254_g2_mul_batch(bn254_ctx *ctx, const bn_g2 *p, const bn_fr *k, bn_g2 *out, size_t n);
/* out[i] = a[i] + b[i]  (negate_b != 0: a[i] - b[i] = a[i] + (-b[i])): `Add`/`Sub` of lib.rs:103-114,146-157 over
   groups/mod.rs:275-347.  The reference's own formulas and branches (zero operands, equal points), so the Jacobian limbs
   returned are the reference's - no normalization involved.  `Neg` is 0 - b. */
int bn254_g1_add_batch(bn254_ctx *ctx, const bn_g1 *a, const bn_g1 *b, bn_g1 *out, size_t n, int negate_b);
int bn254_g2_add_batch(bn254_ctx *ctx, const bn_g2 *a, const bn_g2 *b, bn_g2 *out, size_t n, int negate_b);
/* prepared-G2 mode: precompute once per Q (must not be infinity), then pair many P against it.  `shared` != 0: ONE coefficient
   set (102 entries) is used for every p[i]; otherwise coeffs holds n sets, set i for p[i]. */
int bn254_g2_precompute(bn254_ctx *ctx, const bn_g2 *q, bn_ell_coeffs *coeffs, size_t n);
int bn254_pairing_prepared_batch(bn254_ctx *ctx, const bn_g1 *p, const bn_ell_coeffs *coeffs, int shared, bn_gt *out, size_t n);
/* NATIVE prepared-G2 mode: the device's own counterpart of the reference's internal G2Precomp (groups/mod.rs:472-483) for callers that pair
   many P against the same Q - a verification key - or re-use a set of Q.  bn254_g2_prepare runs precompute (groups/mod.rs:557-588) ONCE per
   point and keeps the result in device memory behind an opaque handle, in the form the kernels consume: the engine's own schedule (6u+2 in
   non-adjacent form: 88 lines instead of 102), every line normalised by its ell_vw coefficient (a factor in Fq2, which the final
   exponentiation kills) and stored as multiplier operands in the 9 x 29-bit limbs of the device arithmetic - 33 792 bytes per point, a
   function of the point alone.  bn254_pairing_prepared_native_batch then computes
       out[i] = final_exponentiation(prepared.miller_loop(p[i]))  =  bn::pairing(p[i], q)          groups/mod.rs:486-519,764-771
   bit-identical to the reference (the Miller VALUE differs by subfield factors; the reference-image coefficients, for the reference's own
   known answers, are bn254_g2_precompute / bn_ell_coeffs above).  A handle made from ONE point is shared by all p[i]; a handle made from nq
   points pairs p[i] with point q_first + i (q_first + n <= nq; the host-buffer entry point uses q_first = 0).  A point at infinity in either
   argument gives Gt::one() (groups/mod.rs:766).  Calls of up to 12 x CUs pairings (3072) are served by the general path's one-pairing-per-wave kernels
   on the points kept with the handle (1.0 ms instead of the 1.7 ms a lane-pair Miller loop needs however few pairings there are; same bytes;
   BN254_OPT_WAVE_PAIRING_MAX = 0 turns that off).  The handle belongs to the context's device; it is immutable after creation, so any
   number of threads / streams may use it concurrently; destroy it after the last call that uses it has completed. */
typedef struct bn254_g2_prepared bn254_g2_prepared;
#define BN254_PREPARED_NATIVE_LINES 88
#define BN254_PREPARED_NATIVE_BYTES 33792    /* device bytes per prepared point: 88 lines x 2 lanes x 12 x 16 B (a handle holds one more record: the identity) */
int bn254_g2_prepare(bn254_ctx *ctx, const bn_g2 *q, size_t nq, bn254_g2_prepared **out);
void bn254_g2_prepared_destroy(bn254_g2_prepared *prep);
size_t bn254_g2_prepared_count(const bn254_g2_prepared *prep);           /* points in the handle */
size_t bn254_g2_prepared_bytes(const bn254_g2_prepared *prep);           /* device memory it holds */
/* copies the table to the host ([line][16-byte group][2 x point + lane] x 4 u32; for tests and inspection): bytes = BN254_PREPARED_NATIVE_BYTES x count */
int bn254_g2_prepared_export(bn254_ctx *ctx, const bn254_g2_prepared *prep, void *host_table, size_t bytes);
int bn254_pairing_prepared_native_batch(bn254_ctx *ctx, const bn_g1 *p, const bn254_g2_prepared *prep, bn_gt *out, size_t n);
/* the multi-pairing over prepared points: out[0] = fold(Gt::one(), acc * pairing(p[i], point i)) over n pairs (shootout/main.rs:11-16 with the
   G2 side prepared; n <= count, or any n against a handle of ONE point) - what a verifier with fixed G2 points evaluates.  ONE final
   exponentiation for the whole product, and from two machine rounds of pairs on (2 x 256 x CUs) two resp. four pairs share one Miller
   accumulator per lane pair: over native tables a pair has no per-step point state, so the shared loop is the line products plus a quarter of
   the squarings.  n == 0 gives Gt::one(); a point at infinity on either side contributes one (groups/mod.rs:766). */
int bn254_pairing_product_prepared_native(bn254_ctx *ctx, const bn_g1 *p, const bn254_g2_prepared *prep, size_t n, bn_gt *out);
int bn254_gt_mul_batch(bn254_ctx *ctx, const bn_gt *a, const bn_gt *b, bn_gt *out, size_t n);
/* Gt::pow (lib.rs:171).  a[i] are Gt VALUES - what the reference's type holds: Gt::one, pairing() and products, powers, inverses of
   such (the Fq12 inside Gt is private and Gt has no decoder), all of order r.  On those the device exponentiates through the
   Frobenius decomposition (k = k0 + k1 q + k2 q^2 + k3 q^3 mod r: 68 cyclotomic squarings and 72 products instead of 252 and 64),
   which needs the order to divide r.  An element that is not even cyclotomic is detected and takes the general chain
   (fields/mod.rs:35-46 as written).  bn254_ctx_set_option(ctx, BN254_OPT_GT_POW_MODE, 2) selects the one-dimensional cyclotomic
   chain, exact for ANY cyclotomic element; 1 the general chain for everything.  One window table of 7.4 KB per lane of a sub-launch
   lives in the context.
   The same precondition holds for bn254_g2_mul_batch: the GLS decomposition multiplies correctly on the order-r subgroup of the twist -
   the only G2 values the reference's API can hold (checked decode, groups/mod.rs:178-205); other twist points are outside the contract. */
int bn254_gt_pow_batch(bn254_ctx *ctx, const bn_gt *a, const bn_fr *k, bn_gt *out, size_t n);
/* out[i] = a[i]^-1 in Fq12 (Gt::inverse, lib.rs:172 -> fields/fq12.rs:284-292); a[i] must be non-zero, as every Gt value is */
int bn254_gt_inverse_batch(bn254_ctx *ctx, const bn_gt *a, bn_gt *out, size_t n);

/* ---- one node, several GPUs (north_star: independent batches shard across the GPUs; ONE exchange for the multi-pairing) --- */
/* `devices[0..ndev)`: HIP device index of every rank (NULL = 0..ndev-1).  One context and one host thread per rank.  A device may
   be listed more than once (several ranks on one GPU - how the N > 1 path is exercised on a one-GPU box); the exchange of the
   product is an RCCL all-gather when all devices are distinct and RCCL loads, peer copies otherwise (bn254_multi_create_ex
   forces one). */
typedef struct bn254_multi bn254_multi;
enum { BN254_EXCHANGE_AUTO = -1, BN254_EXCHANGE_PEER = 0, BN254_EXCHANGE_RCCL = 1 };
int bn254_multi_create(const int *devices, int ndev, bn254_multi **out);                       /* = _ex(..., BN254_EXCHANGE_AUTO, ...) */
/* exchange: BN254_EXCHANGE_AUTO (RCCL when every rank has its own GPU and RCCL loads, else peer copies), _PEER (never load RCCL),
   _RCCL (fail with BN254_E_COMM instead of falling back) */
int bn254_multi_create_ex(const int *devices, int ndev, int exchange, bn254_multi **out);
/* bn254_ctx_set_option on every rank's context */
int bn254_multi_set_option(bn254_multi *m, int key, long value);
void bn254_multi_destroy(bn254_multi *m);
int bn254_multi_device_count(const bn254_multi *m);
int bn254_multi_exchange_kind(const bn254_multi *m);                 /* BN254_EXCHANGE_* */
/* The host thread that drives a rank (its pageable H2D / D2H copies and launches) is pinned, for the duration of a call, to the CPUs
   of that GPU's NUMA node when the node is known (/sys/bus/pci/devices/<bus id>/numa_node) and the process may run there; the
   caller's own thread is never re-pinned (every rank runs on a worker thread of the call).  Returns that node, or -1 when the rank's thread is not pinned. */
int bn254_multi_rank_numa_node(const bn254_multi *m, int rank);
bn254_ctx *bn254_multi_ctx(bn254_multi *m, int rank);                /* rank's context (owned by m) */
/* out[i] = pairing(p[i], q[i]); rank g owns the contiguous shard [n*g/G, n*(g+1)/G); no exchange (BASELINE configs[2]) */
int bn254_pairing_batch_multi(bn254_multi *m, const bn_g1 *p, const bn_g2 *q, bn_gt *out, size_t n);
/* fold(Gt::one(), acc * pairing(p, q)) over all n pairs (shootout/main.rs:11-16): every rank reduces its shard to one
   un-exponentiated Fq12, ONE all-gather of 384 bytes per rank, world-1 products and a single final exponentiation on rank 0
   (BASELINE configs[3]).  Bit-identical to the fold: the final exponentiation is a homomorphism and Gt values are canonical. */
int bn254_pairing_product_multi(bn254_multi *m, const bn_g1 *p, const bn_g2 *q, size_t n, bn_gt *out);

/* native prepared-G2 mode over the GPUs of the handle.  ONE point (nq == 1) is prepared on every rank's GPU and n pairings shard like
   bn254_pairing_batch_multi; nq > 1 points are sharded by the same rule ([nq*g/G, nq*(g+1)/G) on rank g) and then pair with exactly n == nq
   points p[i] (point i with p[i]), so that tables and inputs of a shard live on the same GPU.  No exchange. */
typedef struct bn254_multi_prepared bn254_multi_prepared;
int bn254_g2_prepare_multi(bn254_multi *m, const bn_g2 *q, size_t nq, bn254_multi_prepared **out);
void bn254_multi_prepared_destroy(bn254_multi_prepared *prep);
size_t bn254_multi_prepared_count(const bn254_multi_prepared *prep);
int bn254_pairing_prepared_native_batch_multi(bn254_multi *m, const bn_g1 *p, const bn254_multi_prepared *prep, bn_gt *out, size_t n);
/* bn254_pairing_product_multi over prepared points (bn254_pairing_product_prepared_native sharded): every rank folds its shard over its own tables,
   then the ONE 384-byte exchange and the single final exponentiation on rank 0.  n == count for a sharded set, any n against one point. */
int bn254_pairing_product_prepared_native_multi(bn254_multi *m, const bn_g1 *p, const bn254_multi_prepared *prep, size_t n, bn_gt *out);

/* wire format of the crate's Encodable/Decodable impls for G1/G2 (groups/mod.rs:143-205, fields/fp.rs:24-36, fields/fq2.rs:31-53,
   arith.rs:100-159), as fixed-size batch records: [tag][x][y] with tag 4 and big-endian canonical coordinates (Fq2 = the 512-bit
   integer c1*q + c0); infinity is tag 0 followed by zero padding (the crate's stream emits the lone byte 0).  Decoding validates
   what the crate validates, in its order, and reports per record: 0 ok, 1 "integer is not less than modulus", 2 "integer not
   less than modulus squared", 3 "invalid leading byte", 4 "point is not on the curve", 5 "point is not in the subgroup" (G2
   only).  A rejected record decodes to G::zero(). */
#define BN254_FR_WIRE_BYTES 32    /* Fr: big-endian canonical integer (fields/fp.rs:24-36); decode status 0 or 1, rejected -> Fr::zero() */
#define BN254_G1_WIRE_BYTES 65
#define BN254_G2_WIRE_BYTES 129
int bn254_fr_encode_batch(bn254_ctx *ctx, const bn_fr *k, uint8_t *out, size_t n);
int bn254_fr_decode_batch(bn254_ctx *ctx, const uint8_t *in, bn_fr *out, int32_t *status, size_t n);
int bn254_g1_encode_batch(bn254_ctx *ctx, const bn_g1 *p, uint8_t *out, size_t n);
int bn254_g2_encode_batch(bn254_ctx *ctx, const bn_g2 *p, uint8_t *out, size_t n);
int bn254_g1_decode_batch(bn254_ctx *ctx, const uint8_t *in, bn_g1 *out, int32_t *status, size_t n);
int bn254_g2_decode_batch(bn254_ctx *ctx, const uint8_t *in, bn_g2 *out, int32_t *status, size_t n);
/* the crate's own byte stream (what bincode/rustc_serialize produce for a sequence of points, groups/mod.rs:143-205): a point at
   infinity is the lone byte 0, a finite point is 4 + coordinates, so records have variable length.  encode: `written` bytes are
   produced (BN254_E_BAD_ARG if `cap` is too small).  decode: up to `max_points` points are parsed from `len` bytes; `count` points
   and `consumed` bytes are reported (a truncated trailing record is left unconsumed); status[i] as for the batch decoders.
   DIFFERENCE from the crate: its Decodable returns Err at the first bad record and the caller's stream stops there
   (groups/mod.rs:165-175); this decoder records the status (a bad tag consumes its one byte, a record that fails a check consumes
   its full length), decodes the record to G::zero() and CONTINUES with the next one.  bn254_ctx_set_option(ctx,
   BN254_OPT_STREAM_STOP_AT_ERROR, 1) selects the crate's behaviour: the call stops with the first bad record (count includes it). */
int bn254_g1_encode_stream(bn254_ctx *ctx, const bn_g1 *p, size_t n, uint8_t *out, size_t cap, size_t *written);
int bn254_g2_encode_stream(bn254_ctx *ctx, const bn_g2 *p, size_t n, uint8_t *out, size_t cap, size_t *written);
int bn254_g1_decode_stream(bn254_ctx *ctx, const uint8_t *in, size_t len, bn_g1 *out, int32_t *status, size_t max_points, size_t *count, size_t *consumed);
int bn254_g2_decode_stream(bn254_ctx *ctx, const uint8_t *in, size_t len, bn_g2 *out, int32_t *status, size_t max_points, size_t *count, size_t *consumed);

/* ---- device-resident entry points (inputs/outputs already in HBM; `stream` is a hipStream_t or NULL) ------------------ */
/* Same layouts (array of structs) in device memory.  Asynchronous on `stream`; the caller synchronises. */
int bn254_pairing_batch_dev(bn254_ctx *ctx, const void *d_p, const void *d_q, void *d_out, size_t n, void *stream);
/* the two halves of a pairing, for the multi-pairing product: Miller loop only (infinity -> one), then final exponentiation */
int bn254_miller_batch_dev(bn254_ctx *ctx, const void *d_p, const void *d_q, void *d_f, size_t n, void *stream);
int bn254_final_exp_batch_dev(bn254_ctx *ctx, const void *d_f, void *d_out, size_t n, void *stream);
/* d_out[0] = product of d_in[0..n) in Fq12 (lib.rs:175-179 semantics; order-independent because Fq12 is commutative) */
int bn254_gt_product_dev(bn254_ctx *ctx, const void *d_in, size_t n, void *d_out, void *stream);
/* d_out[0] = final_exponentiation(d_in[0] * ... * d_in[m-1]), m >= 1: the tail of a sharded multi-pairing - the ranks' partial
   products after their exchange, then the ONE final exponentiation (fq12.rs:41-88 behind the fold of shootout/main.rs:11-16).
   Up to 16 values it is a single wave-cooperative launch (one Fq12 spread over a wave, ~0.5 ms instead of 2.9 ms); more go through
   the one-launch product tree first. */
int bn254_gt_product_final_exp_dev(bn254_ctx *ctx, const void *d_in, size_t m, void *d_out, void *stream);
/* local part of a sharded multi-pairing: un-exponentiated product of the Miller values of n pairs -> one Fq12 */
int bn254_miller_product_dev(bn254_ctx *ctx, const void *d_p, const void *d_q, size_t n, void *d_partial, void *stream);
int bn254_g2_precompute_dev(bn254_ctx *ctx, const void *d_q, void *d_coeffs, size_t n, void *stream);
int bn254_miller_prepared_dev(bn254_ctx *ctx, const void *d_p, const void *d_coeffs, int shared, void *d_f, size_t n, void *stream);
/* native prepared-G2 mode on device-resident inputs.  bn254_g2_prepare_dev allocates the handle's table (that part synchronises with the
   device) and enqueues the precompute kernel on `stream`; the other two are asynchronous like the rest of this section.
   bn254_miller_prepared_native_dev returns the un-exponentiated Miller values (only meaningful in front of a final exponentiation). */
int bn254_g2_prepare_dev(bn254_ctx *ctx, const void *d_q, size_t nq, bn254_g2_prepared **out, void *stream);
int bn254_miller_prepared_native_dev(bn254_ctx *ctx, const void *d_p, const bn254_g2_prepared *prep, size_t q_first, void *d_f, size_t n, void *stream);
int bn254_pairing_prepared_native_batch_dev(bn254_ctx *ctx, const void *d_p, const bn254_g2_prepared *prep, size_t q_first, void *d_out, size_t n, void *stream);
/* local part of a multi-pairing over prepared points: un-exponentiated product of the Miller values of p[i] against point q_first + i -> one
   Fq12 (the counterpart of bn254_miller_product_dev; bn254_gt_product_final_exp_dev or bn254_final_exp_batch_dev finishes it) */
int bn254_miller_product_prepared_native_dev(bn254_ctx *ctx, const void *d_p, const bn254_g2_prepared *prep, size_t q_first, size_t n, void *d_partial, void *stream);
int bn254_gt_mul_batch_dev(bn254_ctx *ctx, const void *d_a, const void *d_b, void *d_out, size_t n, void *stream);
int bn254_gt_pow_batch_dev(bn254_ctx *ctx, const void *d_a, const void *d_k, void *d_out, size_t n, void *stream);
int bn254_gt_inverse_batch_dev(bn254_ctx *ctx, const void *d_a, void *d_out, size_t n, void *stream);
int bn254_g1_mul_batch_dev(bn254_ctx *ctx, const void *d_p, const void *d_k, void *d_out, size_t n, void *stream);
int bn254_g2_mul_batch_dev(bn254_ctx *ctx, const void *d_p, const void *d_k, void *d_out, size_t n, void *stream);
/* raw Jacobian result of the reference's MSB-first double-and-add (what G::random produces, groups/mod.rs:220-222):
   used to generate benchmark inputs with z != 1 on the device */
int bn254_g1_mul_jacobian_dev(bn254_ctx *ctx, const void *d_p, const void *d_k, void *d_out, size_t n, void *stream);
int bn254_g2_mul_jacobian_dev(bn254_ctx *ctx, const void *d_p, const void *d_k, void *d_out, size_t n, void *stream);

/* synthetic benchmark inputs, generated in HBM (mirrors benches/api.rs: G::random = one * Fr::random):
   d_out[j] = Montgomery image of (the 512-bit SplitMix64 draw of stream 2*(lo+j)+which, seeded `seed`) mod r - the distribution of
   arith.rs:195-198; equal to bn_amd.distributed.synthetic_scalars word for word.  bn254_tile_dev repeats one record n times. */
int bn254_synthetic_scalars_dev(bn254_ctx *ctx, uint64_t seed, uint64_t lo, size_t n, int which, void *d_out, void *stream);
int bn254_tile_dev(bn254_ctx *ctx, const void *d_record, size_t record_bytes, size_t n, void *d_out, void *stream);

/* ---- measurement ----------------------------------------------------------------------------------------------------- */
/* When enabled, every kernel launch is bracketed by hipEvents on its own stream; bn254_kernel_stats then reports the
   accumulated duration and launch count per kernel since the last reset (this is what bench.py's roofline uses). */
int bn254_profile_enable(bn254_ctx *ctx, int on);
int bn254_profile_reset(bn254_ctx *ctx);
/* kernel: "miller", "miller_shared", "miller_wave", "miller_quad", "pairing_wave", "final_exp", "final_exp_wave", "final_exp_quad", "exp_by_neg_z", "gt_product", "gt_tail", "g1_mul", "g2_mul", "gt_mul", "gt_pow", "g2_precompute", "miller_prepared", "g2_prepare_native", "miller_native", "miller_native_shared", "wire_encode", "wire_decode", "gt_inverse", "g1_add", "g2_add".
   Synchronises and consumes the recorded events (totals accumulate until bn254_profile_reset). */
int bn254_kernel_stats(bn254_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches);
/* issue-rate ceiling of v_mad_u64_u32 (the 32x32+64 multiply-accumulate every field product is built from) at
   `waves_per_simd` resident waves: G lane-MACs per second over the whole chip and the kernel's duration.  bench.py prints it
   as the same-run `roofline.peak`. */
int bn254_ubench_mac32(bn254_ctx *ctx, int waves_per_simd, int iters, double *gmac_per_s, double *ms);
/* the same on operands of `operand_bits` random bits (1..32): the multiplier's rate depends on its data (profiles/r04_ubench_mad_data_dependence.txt);
   29 = the engine's own limbs, which is what bench.py prices `roofline.peak_at_kernel_occupancy` with */
int bn254_ubench_mac32_ex(bn254_ctx *ctx, int waves_per_simd, int iters, int operand_bits, double *gmac_per_s, double *ms);
/* d_out[i] = d_in[i].exp_by_neg_z() as the reference writes it (fields/fq12.rs:229-246), for ANY Fq12: the one function of the path
   whose known answer (fields/mod.rs:171-201) lies OFF the cyclotomic subgroup, where the result depends on the operation sequence.
   The engine's own exponentiation by u (shorter signed-digit chain, equal on every value a pairing produces) is not reachable with
   such an input; this entry point runs the reference's sequence so that its test vector can be checked on the device. */
int bn254_exp_by_neg_z_dev(bn254_ctx *ctx, const void *d_in, void *d_out, size_t n, void *stream);
/* the wave-cooperative machine on one wave: milliseconds for `iters` runs of program `which` (0 cyclotomic squaring, 1 Fq12
   product, 2 slot copy, 3 Frobenius map, 4 whole final exponentiation, 5 a fused run of five squarings) - the per-phase costs quoted in DESIGN.md */
int bn254_wave_ubench(bn254_ctx *ctx, int which, int iters, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* BN254_HIP_H */

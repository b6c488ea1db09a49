"""CPU-side proof of the DEVICE arithmetic: bn_amd/csrc/*.hpp compiled with g++ (tests/hostsim/hostsim.cpp, -DBN_BOUNDS
so every limb/value bound of the lazy 9x29-bit number system is enforced at run time) and compared bit for bit with the
oracle.  Not a fallback: the product never loads this library."""
import numpy as np
import pytest

import bn_model as M
import hostsim_lib
from conftest import canon_infinity
from bn_oracle import FQ, FR


@pytest.fixture(scope="module")
def hs():
    return hostsim_lib.HostSim(bounds=True)


def _rfq(oracle, rng):
    return oracle.fp_from_int(FQ, int.from_bytes(rng.bytes(40), "little") % M.Q)


def _rf(oracle, rng, n):
    return np.concatenate([_rfq(oracle, rng) for _ in range(n)])


EDGE = [0, 1, 2, M.Q - 1, M.Q - 2, (M.Q - 1) // 2, (1 << 253), (1 << 29) - 1, 1 << 232, M.MONT_R % M.Q]


def test_fe_ops_match_oracle(oracle, hs):
    rng = np.random.default_rng(11)
    vals = [oracle.fp_from_int(FQ, v) for v in EDGE] + [_rfq(oracle, rng) for _ in range(60)]
    for i, a in enumerate(vals):
        assert np.array_equal(hs.call("hs_fe_roundtrip", a, out_words=8), a)
        for b in vals[::7] + vals[:len(EDGE)]:
            assert np.array_equal(hs.call("hs_fe_mul", a, b, out_words=8), oracle.fp_mul(FQ, a, b))
            assert np.array_equal(hs.call("hs_fe_add", a, b, out_words=8), oracle.fp_add(FQ, a, b))
            assert np.array_equal(hs.call("hs_fe_sub", a, b, out_words=8), oracle.fp_sub(FQ, a, b))
        assert bool(hs.lib.hs_fe_is_zero(a.ctypes.data_as(hostsim_lib._U32P))) == (i == 0)


def test_fe_lazy_forms(oracle, hs):
    rng = np.random.default_rng(12)
    pool = [oracle.fp_from_int(FQ, v) for v in EDGE] + [_rfq(oracle, rng) for _ in range(40)]
    for _ in range(400):
        a, b, c, d = (pool[i] for i in rng.integers(0, len(pool), 4))
        ai, bi, ci = (oracle.fp_to_int(FQ, x) for x in (a, b, c))
        want = oracle.fp_from_int(FQ, (17 * ai + 18 * bi - 6 * ci) % M.Q)
        assert np.array_equal(hs.call("hs_fe_lazy_mix", a, b, c, out_words=8), want)
        want = oracle.fp_from_int(FQ, (10 * ai - 10 * bi - 8 * ci) % M.Q)
        assert np.array_equal(hs.call("hs_fe_signed_mix", a, b, c, out_words=8), want)
        want = oracle.fp_add(FQ, oracle.fp_mul(FQ, a, b), oracle.fp_mul(FQ, c, d))
        assert np.array_equal(hs.call("hs_fe_mul2", a, b, c, d, out_words=8), want)
        # the signed dual product (Karatsuba cross products): operands are differences of standard elements, the result a signed lazy value
        sub, mul = (lambda x, y: oracle.fp_sub(FQ, x, y)), (lambda x, y: oracle.fp_mul(FQ, x, y))
        want = oracle.fp_add(FQ, mul(sub(a, b), sub(c, d)), mul(sub(b, a), sub(d, a)))
        assert np.array_equal(hs.call("hs_fe_mul2s", a, b, c, d, out_words=8), want)


def test_fe_inverse(oracle, hs):
    """divsteps (safegcd) inversion == the reference's binary EEA result == the Fermat chain, incl. edge values"""
    rng = np.random.default_rng(13)
    for a in [oracle.fp_from_int(FQ, v) for v in EDGE[1:] + [3, 12345, (1 << 254) % M.Q, M.Q - 3]] + [_rfq(oracle, rng) for _ in range(300)]:
        want = oracle.fp_inverse(FQ, a)
        assert np.array_equal(hs.call("hs_fe_inverse", a, out_words=8), want)
    for a in [_rfq(oracle, rng) for _ in range(3)]:
        assert np.array_equal(hs.call("hs_fe_inverse_fermat", a, out_words=8), oracle.fp_inverse(FQ, a))
    z = oracle.fp_from_int(FQ, 0)
    assert np.array_equal(hs.call("hs_fe_inverse", z, out_words=8), z)       # engine convention: inverse(0) = 0


def test_fq2(oracle, hs):
    rng = np.random.default_rng(14)
    for _ in range(25):
        a, b = _rf(oracle, rng, 2), _rf(oracle, rng, 2)
        assert np.array_equal(hs.call("hs_fq2_mul", a, b, out_words=16), oracle.fq2_mul(a, b))
        assert np.array_equal(hs.call("hs_fq2_sqr", a, out_words=16), oracle.fq2_sqr(a))
        assert np.array_equal(hs.call("hs_fq2_mul_xi", a, out_words=16), oracle.fq2_mul_xi(a))
        assert np.array_equal(hs.call("hs_fq2_inverse", a, out_words=16), oracle.fq2_inverse(a))


def test_fq12_ops(oracle, hs, kats):
    rng = np.random.default_rng(15)
    elems = [oracle.fq12_from_ints(kats["fq12_test_vector"]["start"]), oracle.fq12_from_ints(kats["test_cyclotomic_exp"]["orig"]),
             oracle.fq12_one()] + [_rf(oracle, rng, 12) for _ in range(4)]
    for a in elems:
        b = elems[int(rng.integers(0, len(elems)))]
        assert np.array_equal(hs.call("hs_fq12_mul", a, b, out_words=96), oracle.fq12_mul(a, b))
        assert np.array_equal(hs.call("hs_fq12_sqr", a, out_words=96), oracle.fq12_sqr(a))
        assert np.array_equal(hs.call("hs_fq12_conj", a, out_words=96), oracle.fq12_unitary_inverse(a))
        assert np.array_equal(hs.call("hs_fq12_cyclotomic_sqr", a, out_words=96), oracle.fq12_cyclotomic_squared(a))
        assert np.array_equal(hs.call("hs_fq12_inverse", a, out_words=96), oracle.fq12_inverse(a))
        for p in (1, 2, 3):
            assert np.array_equal(hs.call("hs_fq12_frobenius", a, p, out_words=96), oracle.fq12_frobenius_map(a, p))
        l = _rf(oracle, rng, 6)
        assert np.array_equal(hs.call("hs_fq12_mul_by_024", a, l[:8], l[8:16], l[16:], out_words=96),
                              oracle.fq12_mul_by_024(a, l[:8], l[8:16], l[16:]))


def test_kats_through_engine(oracle, hs, kats):
    """the reference's own known answers, computed by the engine's code (fields/mod.rs:171-201, groups/mod.rs:522-547,773-796)"""
    I = lambda l: [int(x) for x in l]
    e = hs.call("hs_fq12_exp_by_neg_z", oracle.fq12_from_ints(kats["test_cyclotomic_exp"]["orig"]), out_words=96)
    assert oracle.fq12_to_ints(e) == I(kats["test_cyclotomic_exp"]["expected"])
    k1 = oracle.fp_from_decimal(FR, kats["test_miller_loop"]["k1"]); k2 = oracle.fp_from_decimal(FR, kats["test_miller_loop"]["k2"])
    P = oracle.g1_mul(oracle.g1_one(), k1); Q = oracle.g2_mul(oracle.g2_one(), k2)
    # the NAF schedule used inside final_exponentiation equals the reference's on the cyclotomic subgroup
    cyc = oracle.fq12_final_exp_first_chunk(oracle.miller_only(P, Q))
    assert np.array_equal(hs.call("hs_fq12_exp_by_neg_z_naf", cyc, out_words=96), oracle.fq12_exp_by_neg_z(cyc))
    assert oracle.fq12_to_ints(hs.call("hs_miller", P, Q, out_words=96)) == I(kats["test_miller_loop"]["expected"])
    assert oracle.fq12_to_ints(hs.call("hs_pairing", P, Q, out_words=96)) == I(kats["test_reduced_pairing"]["expected"])


def test_pairing_random_and_edges(oracle, hs):
    rng = np.random.default_rng(16)
    scal = [1, 2, M.R_ORD - 1, 3, (1 << 253) % M.R_ORD] + [int.from_bytes(rng.bytes(40), "little") % M.R_ORD for _ in range(6)]
    for i in range(len(scal)):
        P = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_int(FR, scal[i]))
        Q = oracle.g2_mul(oracle.g2_one(), oracle.fp_from_int(FR, scal[-1 - i]))
        got = hs.call("hs_pairing", P, Q, out_words=96)
        assert np.array_equal(got, oracle.pairing(P, Q))
    # z == 1 inputs (the reference's shortcut, groups/mod.rs:116-120) and infinity (-> one, :766)
    got = hs.call("hs_pairing", oracle.g1_one(), oracle.g2_one(), out_words=96)
    assert np.array_equal(got, oracle.pairing(oracle.g1_one(), oracle.g2_one()))
    one = oracle.fq12_one()
    assert np.array_equal(hs.call("hs_pairing", oracle.g1_zero(), oracle.g2_one(), out_words=96), one)
    assert np.array_equal(hs.call("hs_pairing", oracle.g1_one(), oracle.g2_zero(), out_words=96), one)


def test_lane_pair_mapping(oracle, hs, kats):
    """Fq2B (one element per lane pair) executed on a simulated lane pair: same bytes as the oracle"""
    rng = np.random.default_rng(17)
    for _ in range(20):
        a, b = _rf(oracle, rng, 2), _rf(oracle, rng, 2)
        assert np.array_equal(hs.call("hsb_fq2_mul", a, b, out_words=16), oracle.fq2_mul(a, b))
        assert np.array_equal(hs.call("hsb_fq2_sqr", a, out_words=16), oracle.fq2_sqr(a))
        assert np.array_equal(hs.call("hsb_fq2_mul_xi", a, out_words=16), oracle.fq2_mul_xi(a))
        assert np.array_equal(hs.call("hsb_fq2_inverse", a, out_words=16), oracle.fq2_inverse(a))
    for a in [oracle.fq12_from_ints(kats["fq12_test_vector"]["start"]), oracle.fq12_one()] + [_rf(oracle, rng, 12) for _ in range(3)]:
        b = _rf(oracle, rng, 12)
        assert np.array_equal(hs.call("hsb_fq12_mul", a, b, out_words=96), oracle.fq12_mul(a, b))
        assert np.array_equal(hs.call("hsb_fq12_sqr", a, out_words=96), oracle.fq12_sqr(a))
        assert np.array_equal(hs.call("hsb_fq12_cyclotomic_sqr", a, out_words=96), oracle.fq12_cyclotomic_squared(a))
        assert np.array_equal(hs.call("hsb_fq12_inverse", a, out_words=96), oracle.fq12_inverse(a))
        for p in (1, 2, 3):
            assert np.array_equal(hs.call("hsb_fq12_frobenius", a, p, out_words=96), oracle.fq12_frobenius_map(a, p))
        l = _rf(oracle, rng, 6)
        assert np.array_equal(hs.call("hsb_fq12_mul_by_024", a, l[:8], l[8:16], l[16:], out_words=96),
                              oracle.fq12_mul_by_024(a, l[:8], l[8:16], l[16:]))
    a = _rf(oracle, rng, 12)
    assert np.array_equal(hs.call("hsb_final_exponentiation", a, out_words=96), oracle.fq12_final_exponentiation(a))
    I = lambda l: [int(x) for x in l]
    k1 = oracle.fp_from_decimal(FR, kats["test_miller_loop"]["k1"]); k2 = oracle.fp_from_decimal(FR, kats["test_miller_loop"]["k2"])
    P = oracle.g1_mul(oracle.g1_one(), k1); Q = oracle.g2_mul(oracle.g2_one(), k2)
    P0, Q0 = P, Q
    assert oracle.fq12_to_ints(hs.call("hsb_pairing", P, Q, out_words=96)) == I(kats["test_reduced_pairing"]["expected"])
    for _ in range(3):
        P = oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng)); Q = oracle.g2_mul(oracle.g2_one(), _fr(oracle, rng))
        assert np.array_equal(hs.call("hsb_pairing", P, Q, out_words=96), oracle.pairing(P, Q))
    assert np.array_equal(hs.call("hsb_pairing", oracle.g1_zero(), Q, out_words=96), oracle.fq12_one())
    assert np.array_equal(hs.call("hsb_pairing", oracle.g1_one(), oracle.g2_one(), out_words=96), oracle.pairing(oracle.g1_one(), oracle.g2_one()))
    # the NAF Miller schedule of the pairing kernels: same pairing value bit for bit (incl. the reference's known answer);
    # the reference-schedule Miller value itself equals the reference's
    assert oracle.fq12_to_ints(hs.call("hsb_pairing_naf", P0, Q0, out_words=96)) == I(kats["test_reduced_pairing"]["expected"])
    assert oracle.fq12_to_ints(hs.call("hsb_miller", P0, Q0, out_words=96)) == I(kats["test_miller_loop"]["expected"])
    for _ in range(3):
        P = oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng)); Q = oracle.g2_mul(oracle.g2_one(), _fr(oracle, rng))
        assert np.array_equal(hs.call("hsb_pairing_naf", P, Q, out_words=96), oracle.pairing(P, Q))
    assert np.array_equal(hs.call("hsb_pairing_naf", oracle.g1_one(), oracle.g2_one(), out_words=96), oracle.pairing(oracle.g1_one(), oracle.g2_one()))
    assert np.array_equal(hs.call("hsb_pairing_naf", P, oracle.g2_zero(), out_words=96), oracle.fq12_one())


def _fr(oracle, rng):
    return oracle.fp_from_int(FR, int.from_bytes(rng.bytes(40), "little") % M.R_ORD)


def test_scalar_mul_reference_chain(oracle, hs):
    """G * Fr through the engine's Jacobian code: the raw coordinates equal the reference's double-and-add chain bit for bit"""
    rng = np.random.default_rng(18)
    ks = [0, 1, 2, 3, M.R_ORD - 1, M.R_ORD - 2, 1 << 200] + [int.from_bytes(rng.bytes(40), "little") % M.R_ORD for _ in range(3)]
    b1 = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_int(FR, 777)); b2 = oracle.g2_mul(oracle.g2_one(), oracle.fp_from_int(FR, 999))
    for kv in ks:
        k = oracle.fp_from_int(FR, kv)
        raw = hs.call("hs_fr_from_mont", k, out_words=8)
        assert sum(int(x) << (64 * i) for i, x in enumerate(raw)) == kv
        for base, fn, w, om, on in ((b1, "hs_g1_mul", 12, oracle.g1_mul, oracle.g1_normalize), (oracle.g1_zero(), "hs_g1_mul", 12, oracle.g1_mul, oracle.g1_normalize),
                                    (b2, "hs_g2_mul", 24, oracle.g2_mul, oracle.g2_normalize), (oracle.g2_one(), "hs_g2_mul", 24, oracle.g2_mul, oracle.g2_normalize),
                                    (b2, "hsb_g2_mul", 24, oracle.g2_mul, oracle.g2_normalize), (oracle.g2_zero(), "hsb_g2_mul", 24, oracle.g2_mul, oracle.g2_normalize)):
            want = om(base, k)
            assert np.array_equal(hs.call(fn, base, k, 0, out_words=2 * w), want)
            assert np.array_equal(hs.call(fn, base, k, 1, out_words=2 * w), canon_infinity(on(want)))
            assert np.array_equal(hs.call(fn, base, k, 2, out_words=2 * w), canon_infinity(on(want)))   # windowed algorithm, normalized
            assert np.array_equal(hs.call(fn, base, k, 3, out_words=2 * w), canon_infinity(on(want)))   # Booth windows, affine table on the isomorphic curve


def test_prepared_mode_and_product_chain(oracle, hs, kats):
    """the prepared-G2 kernels' code path (precompute_lines -> stored coefficients -> miller_loop_prepared -> final exponentiation) and
    the product chain of the multi-pairing tree, with bound enforcement: coefficients equal the oracle's precompute, the pairing
    equals pairing(), the chain equals the oracle's fold"""
    rng = np.random.default_rng(31)
    P = oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng)); Q = oracle.g2_mul(oracle.g2_one(), _fr(oracle, rng))
    coeffs = np.zeros(102 * 24, np.uint64)
    lib = hs.lib
    import ctypes as C
    U32 = C.POINTER(C.c_uint32)
    out = np.zeros(48, np.uint64)
    lib.hsb_prepared_pairing(P.ctypes.data_as(U32), Q.ctypes.data_as(U32), coeffs.ctypes.data_as(U32), out.ctypes.data_as(U32))
    assert np.array_equal(coeffs.reshape(102, 3, 8), oracle.g2_precompute(oracle.g2_to_affine(Q)))
    assert np.array_equal(out, oracle.pairing(P, Q))
    vals = np.stack([oracle.pairing(oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng)), Q) for _ in range(5)] + [oracle.miller_only(P, Q)])
    want = vals[0]
    for v in vals[1:]:
        want = oracle.fq12_mul(want, v)
    got = np.zeros(48, np.uint64)
    lib.hsb_gt_product(np.ascontiguousarray(vals).ctypes.data_as(U32), C.c_int(len(vals)), got.ctypes.data_as(U32))
    assert np.array_equal(got, want)


def test_native_prepared_mode(oracle, hs, kats):
    """the NATIVE prepared-G2 kernels' code path (pairing.hpp precompute_native -> table -> miller_loop_native -> final exponentiation) with
    every limb / value bound enforced: the reference's known answer (groups/mod.rs:773-796) through the table of its Q, random P incl. the
    generator with z = 1 and infinity against the oracle's pairing(), the table's records canonical, and two Jacobian representations of one
    Q giving the same table"""
    import ctypes as C
    U32 = C.POINTER(C.c_uint32)
    lib = hs.lib
    assert lib.hsb_native_lines() == 88
    k = kats["test_reduced_pairing"]
    P = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_decimal(FR, k["k1"])); Q = oracle.g2_mul(oracle.g2_one(), oracle.fp_from_decimal(FR, k["k2"]))
    tab = np.zeros((88, 2, 48), np.uint32); out = np.zeros(48, np.uint64)
    lib.hsb_native_precompute(Q.ctypes.data_as(U32), tab.ctypes.data_as(U32))
    lib.hsb_native_pairing(P.ctypes.data_as(U32), tab.ctypes.data_as(U32), C.c_int(0), out.ctypes.data_as(U32))
    assert oracle.fq12_to_ints(out) == [int(x) for x in k["expected"]]
    tab2 = np.zeros_like(tab)
    lib.hsb_native_precompute(oracle.g2_normalize(Q).ctypes.data_as(U32), tab2.ctypes.data_as(U32))
    assert np.array_equal(tab, tab2)
    for off in (0, 9, 18, 28, 37):                     # every stored operand is a canonical field element in 29-bit limbs
        limbs = tab[:, :, off:off + 9].astype(object)
        vals = sum(limbs[:, :, i] * (1 << (29 * i)) for i in range(9))
        assert (vals < M.Q).all() and (tab[:, :, off:off + 8] < (1 << 29)).all()
    rng = np.random.default_rng(33)
    for Pi in (oracle.g1_one(), oracle.g1_zero(), oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng)), oracle.g1_mul(oracle.g1_one(), oracle.fp_from_int(FR, M.R_ORD - 1))):
        lib.hsb_native_pairing(np.ascontiguousarray(Pi).ctypes.data_as(U32), tab.ctypes.data_as(U32), C.c_int(0), out.ctypes.data_as(U32))
        assert np.array_equal(out, oracle.pairing(Pi, Q))


def test_native_prepared_product(oracle, hs):
    """the multi-pairing over native tables (pairing.hpp miller_loop_native_shared: M pairs of one lane pair share the accumulator; sigma, tau
    per pair, 9 tau / -+tau re-derived per line; infinity = the identity record with sigma = 1, tau = 0), every bound enforced: M = 1 ... 4
    against the oracle's fold of pairing() values (shootout/main.rs:11-16), with P or Q at infinity in any slot"""
    import ctypes as C
    U32 = C.POINTER(C.c_uint32)
    lib = hs.lib
    rng = np.random.default_rng(41)
    Ps = [oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng)) for _ in range(4)]
    Qs = [oracle.g2_mul(oracle.g2_one(), _fr(oracle, rng)) for _ in range(4)]
    tabs = np.zeros((4, 88, 2, 48), np.uint32)
    for i, Q in enumerate(Qs):
        lib.hsb_native_precompute(Q.ctypes.data_as(U32), tabs[i].ctypes.data_as(U32))
    vals = [oracle.pairing(P, Q) for P, Q in zip(Ps, Qs)]
    one = oracle.pairing(oracle.g1_zero(), Qs[0])

    def run(m, p_inf=(), q_inf=()):
        g1 = np.stack([oracle.g1_zero() if i in p_inf else Ps[i] for i in range(m)])
        flags = np.array([1 if i in q_inf else 0 for i in range(m)], np.uint32)
        out = np.zeros(48, np.uint64)
        lib.hsb_native_product(C.c_int(m), np.ascontiguousarray(g1).ctypes.data_as(U32), tabs.ctypes.data_as(U32), flags.ctypes.data_as(U32), C.c_int(0), out.ctypes.data_as(U32))
        want = one
        for i in range(m):
            if i not in p_inf and i not in q_inf:
                want = oracle.fq12_mul(want, vals[i])
        assert np.array_equal(out, want), (m, p_inf, q_inf)

    for m in (1, 2, 3, 4):
        run(m)
    run(4, p_inf=(0,)); run(4, q_inf=(3,)); run(4, p_inf=(1,), q_inf=(2,)); run(2, p_inf=(0, 1)); run(3, p_inf=(2,), q_inf=(0, 1))


def test_gt_pow_windowed_chain(oracle, hs):
    """Gt::pow as bn254_gt_pow_B computes it (4-bit windows, general squarings), every limb/value bound enforced: pairing values, an
    element outside the cyclotomic subgroup (a raw Miller value) and edge exponents against the oracle's bit-serial pow"""
    rng = np.random.default_rng(29)
    P = oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng)); Q = oracle.g2_mul(oracle.g2_one(), _fr(oracle, rng))
    g = oracle.pairing(P, Q); raw = oracle.miller_only(P, Q)
    for kv in (0, 1, 2, 15, 16, M.R_ORD - 1, (1 << 253) + 12345, int.from_bytes(rng.bytes(40), "little") % M.R_ORD):
        k = oracle.fp_from_int(FR, kv)
        for a in (g, raw):
            assert np.array_equal(hs.call("hsb_gt_pow", a, k, out_words=96), oracle.gt_pow(a, k)), kv


def test_g1_glv_scalar_mul(oracle, hs, ref_consts):
    """bn254_g1_mul_batch's chain: k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^129, Booth radix-16 digits, phi(x, y) = (beta x, y);
    the normalized result equals the reference's G * Fr for edge and random scalars and base points (incl. infinity)"""
    import re, pathlib
    txt = (pathlib.Path(__file__).resolve().parents[1] / "bn_amd" / "csrc" / "bn254_constants.hpp").read_text()
    lam = sum(int(x, 16) << (32 * i) for i, x in enumerate(re.findall(r"0x[0-9a-f]+", re.search(r"GLV_LAMBDA\[8\] = \{(.*?)\}", txt).group(1))))
    assert (lam * lam + lam + 1) % M.R_ORD == 0
    rng = np.random.default_rng(23)
    ks = [0, 1, 2, 15, 16, 17, M.R_ORD - 1, M.R_ORD - 2, lam, lam - 1, M.R_ORD - lam, 1 << 127, (1 << 128) - 1, 1 << 253] + \
         [int.from_bytes(rng.bytes(40), "little") % M.R_ORD for _ in range(12)]
    b1 = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_int(FR, 31337))
    for kv in ks:
        k = oracle.fp_from_int(FR, kv)
        d = hs.call("hs_glv_decompose", k, out_words=12).view(np.uint32)
        m1 = sum(int(d[i]) << (32 * i) for i in range(5)); m2 = sum(int(d[6 + i]) << (32 * i) for i in range(5))
        assert m1 < (1 << 129) and m2 < (1 << 129)
        assert ((-m1 if d[5] else m1) + (-m2 if d[11] else m2) * lam - kv) % M.R_ORD == 0
        for base in (b1, oracle.g1_one(), oracle.g1_zero()):
            want = canon_infinity(oracle.g1_normalize(oracle.g1_mul(base, k)))
            assert np.array_equal(hs.call("hs_g1_mul_glv", base, k, out_words=24), want), kv
    # P and -P style cancellations inside the interleaved chain: k1 P + k2 phi(P) where k = lambda (k1 = 0, k2 = 1) and k = r - lambda
    for kv in (lam, M.R_ORD - lam, (lam + 1) % M.R_ORD):
        k = oracle.fp_from_int(FR, kv)
        assert np.array_equal(hs.call("hs_g1_mul_glv", b1, k, out_words=24), canon_infinity(oracle.g1_normalize(oracle.g1_mul(b1, k))))


def test_many_random_pairings_with_bound_verification(oracle, hs):
    """200 random pairings through BOTH lane mappings of the device code on the CPU, every operation checked against its claimed
    limb/value bounds (actual limbs, not just worst cases) and every result against the oracle"""
    rng = np.random.default_rng(19)
    n = 100
    k1 = np.stack([_fr(oracle, rng) for _ in range(n)]); k2 = np.stack([_fr(oracle, rng) for _ in range(n)])
    P = oracle.g1_mul_batch_jacobian(np.tile(oracle.g1_one(), (n, 1)), k1, 1)
    Q = oracle.g2_mul_batch_jacobian(np.tile(oracle.g2_one(), (n, 1)), k2, 1)
    want = oracle.pairing_batch(P, Q, 1)
    for i in range(n):
        assert np.array_equal(hs.call("hs_pairing", P[i], Q[i], out_words=96), want[i])
        assert np.array_equal(hs.call("hsb_pairing", P[i], Q[i], out_words=96), want[i])


def test_wire_format_through_engine_code(oracle, hs):
    """encode/decode records and every validation outcome, device code on the CPU vs the code-derived oracle"""
    import ctypes as C
    from conftest import g2_point_outside_subgroup
    def enc(fn, p, n):
        out = np.zeros(n, np.uint8); p = np.ascontiguousarray(p, np.uint64)
        getattr(hs.lib, fn)(p.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)); return out
    def dec(fn, b, w):
        out = np.zeros(w, np.uint64); b = np.ascontiguousarray(b, np.uint8)
        rc = getattr(hs.lib, fn)(b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)); return rc, out
    rng = np.random.default_rng(23)
    for _ in range(4):
        k = _fr(oracle, rng)
        P = oracle.g1_mul(oracle.g1_one(), k); Q = oracle.g2_mul(oracle.g2_one(), k)
        e1 = enc("hs_g1_encode", P, 65); e2 = enc("hs_g2_encode", Q, 129)
        assert np.array_equal(e1, oracle.g1_encode(P)) and np.array_equal(e2, oracle.g2_encode(Q))
        rc, d = dec("hs_g1_decode", e1, 12); assert rc == 0 and np.array_equal(d, oracle.g1_normalize(P))
        rc, d = dec("hs_g2_decode", e2, 24); assert rc == 0 and np.array_equal(d, oracle.g2_normalize(Q))
        for pos in (0, 5, 40):
            b = e1.copy(); b[pos] ^= 0x55
            assert dec("hs_g1_decode", b, 12)[0] == oracle.g1_decode(b)[0] != 0
        for pos in (0, 5, 70, 128):
            b = e2.copy(); b[pos] ^= 0x55
            assert dec("hs_g2_decode", b, 24)[0] == oracle.g2_decode(b)[0] != 0
    assert np.array_equal(enc("hs_g1_encode", oracle.g1_zero(), 65), oracle.g1_encode(oracle.g1_zero()))
    rc, d = dec("hs_g2_decode", oracle.g2_encode(oracle.g2_zero()), 24); assert rc == 0 and np.array_equal(d, oracle.g2_zero())
    b = oracle.g1_encode(P).copy(); b[1:33] = 255; assert dec("hs_g1_decode", b, 12)[0] == oracle.g1_decode(b)[0] == 1
    b = oracle.g2_encode(Q).copy(); b[1:65] = 255; assert dec("hs_g2_decode", b, 24)[0] == oracle.g2_decode(b)[0] == 2
    bad = g2_point_outside_subgroup()
    assert oracle.g2_decode(bad)[0] == 5 and dec("hs_g2_decode", bad, 24)[0] == 5
    # Fr records (fields/fp.rs:24-36): canonical big-endian integer, rejected when >= r
    for v in (0, 1, M.R_ORD - 1, 1 << 253, int.from_bytes(rng.bytes(40), "little") % M.R_ORD):
        k = oracle.fp_from_int(FR, v)
        e = enc("hs_fr_encode", k, 32)
        assert bytes(e) == v.to_bytes(32, "big") == bytes(oracle.fr_encode(k))
        rc, d = dec("hs_fr_decode", e, 4); assert rc == 0 and np.array_equal(d, k)
    for v in (M.R_ORD, M.R_ORD + 5, (1 << 256) - 1):
        b = np.frombuffer(v.to_bytes(32, "big"), np.uint8)
        rc, d = dec("hs_fr_decode", b, 4); assert rc == 1 == oracle.fr_decode(b)[0] and not d.any()


def test_golden_fixtures_through_engine_code(hs, goldens):
    """committed fixtures (edge scalars first) through the device code on the CPU, both Miller schedules"""
    g = goldens
    for i in (0, 1, 2, 3, 10, 50):
        assert np.array_equal(hs.call("hsb_pairing_naf", g["g1"][i], g["g2"][i], out_words=96), g["gt"][i])
    assert np.array_equal(hs.call("hsb_pairing", g["g1"][2], g["g2"][2], out_words=96), g["gt"][2])


def test_group_addition_branches(oracle, hs):
    """groups/mod.rs:275-347 through the engine code: generic sum, a + a (doubling branch), a + 0, 0 + b, a - a, 0 - b, 0 - 0:
    the raw Jacobian limbs equal the reference's in every branch"""
    rng = np.random.default_rng(29)
    a1 = oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng)); b1 = oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng))
    a2 = oracle.g2_mul(oracle.g2_one(), _fr(oracle, rng)); b2 = oracle.g2_mul(oracle.g2_one(), _fr(oracle, rng))
    for fn, w, a, b, z, add, neg in (("hs_g1_add", 12, a1, b1, oracle.g1_zero(), oracle.g1_add, oracle.g1_neg),
                                      ("hsb_g2_add", 24, a2, b2, oracle.g2_zero(), oracle.g2_add, oracle.g2_neg)):
        for x, y in ((a, b), (a, a), (a, z), (z, b), (z, z), (b, a)):
            assert np.array_equal(hs.call(fn, x, y, 0, out_words=2 * w), add(x, y))
            assert np.array_equal(hs.call(fn, x, y, 1, out_words=2 * w), add(x, neg(y)))


def executed_chain_lengths(oracle, hs):
    """(Fq-product equivalents, multiply instructions) per unit of the chains the kernels execute, counted by the host simulation of the
    device code on one unit (the control flow is data independent).  Fq-product equivalents: fe_mul + 1.5 x fe_mul2, both lanes of a lane
    pair - the scale of `roofline.frac` of the side kernels.  Multiply instructions: what the GPU leaves issue (v_mad_u64_u32, v_mad_i64_i32,
    v_mul_lo, v_mul_hi: 171 per product, 252 per dual product, 576 / 495 per lazily reduced chain of six / five, the fused reductions' own),
    PER LANE - the scale of `roofline.frac_executed`"""
    import ctypes as C
    rng = np.random.default_rng(77)
    k1, k2 = (oracle.fp_from_int(FR, int.from_bytes(rng.bytes(40), "little") % M.R_ORD) for _ in range(2))
    P = oracle.g1_mul(oracle.g1_one(), k1); Q = oracle.g2_mul(oracle.g2_one(), k1)
    g = oracle.pairing(P, Q)
    U32 = C.POINTER(C.c_uint32)
    prods, macs = {}, {}
    def count_raw(name, lanes, fn, *args):
        hs.lib.hs_counts_reset()
        fn(*args)
        a = (C.c_ulong * 16)(); hs.lib.hs_counts_get(a)
        prods[name] = int(a[0] + 1.5 * a[1]); macs[name] = int(a[8]) // lanes
    def count(name, lanes, fn, *args, out_words):
        count_raw(name, lanes, lambda: hs.call(fn, *args, out_words=out_words))
    count("g1_mul", 1, "hs_g1_mul_glv", P, k2, out_words=24)
    count("g2_mul", 2, "hsb_g2_mul_gls", Q, k2, out_words=48)
    count("gt_pow", 2, "hsb_gt_pow_auto", g, k2, out_words=96)                   # a pairing value: membership test + cyclotomic chain
    # the headline kernels: the fused NAF Miller loop (prologue included) and the final exponentiation
    count("miller", 2, "hsb_miller_naf", P, Q, out_words=96)
    count("final_exp", 2, "hsb_final_exponentiation", oracle.miller_only(P, Q), out_words=96)
    # prepared-G2 mode: the Miller kernels alone, over the native table resp. the reference-image coefficients
    tab = np.zeros(88 * 2 * 48, np.uint32); coeffs = np.zeros(102 * 24, np.uint64); o = np.zeros(48, np.uint64)
    count_raw("g2_prepare_native", 2, hs.lib.hsb_native_precompute, Q.ctypes.data_as(U32), tab.ctypes.data_as(U32))
    count_raw("miller_native", 2, hs.lib.hsb_native_pairing, P.ctypes.data_as(U32), tab.ctypes.data_as(U32), C.c_int(1), o.ctypes.data_as(U32))
    hs.lib.hsb_prepared_pairing(P.ctypes.data_as(U32), Q.ctypes.data_as(U32), coeffs.ctypes.data_as(U32), o.ctypes.data_as(U32))
    count_raw("miller_prepared", 2, hs.lib.hsb_prepared_miller, P.ctypes.data_as(U32), coeffs.ctypes.data_as(U32), o.ctypes.data_as(U32))
    # the multi-pairing over native tables: PER PAIR of a lane pair that carries four (resp. two) pairs on one accumulator
    P4 = np.ascontiguousarray(np.tile(P, (4, 1))); tab4 = np.ascontiguousarray(np.tile(tab, 4)); flags = np.zeros(4, np.uint32)
    for m in (4, 2):
        count_raw(f"miller_native_shared{m}", 2, hs.lib.hsb_native_product, C.c_int(m), P4.ctypes.data_as(U32), tab4.ctypes.data_as(U32), flags.ctypes.data_as(U32), C.c_int(1), o.ctypes.data_as(U32))
        prods[f"miller_native_shared{m}"] //= m; macs[f"miller_native_shared{m}"] //= m
    return prods, macs


def test_executed_chain_lengths(oracle, hs):
    """bench.py prices `roofline.frac` of the side kernels over the chain they EXECUTE and `roofline.frac_executed` of every kernel over
    the multiply instructions it issues: the committed figures are the simulation's"""
    import json, pathlib
    want = json.loads((pathlib.Path(__file__).resolve().parents[1] / "profiles" / "executed_chain_lengths.json").read_text())
    prods, macs = executed_chain_lengths(oracle, hs)
    assert prods == want["fq_products_per_unit"]
    assert macs == want["mac_instructions_per_lane_and_unit"]


def test_gt_pow_cyclotomic_chain(oracle, hs):
    """bn254_gt_pow_B's fast path: the device's membership test separates pairing values (and one) from arbitrary Fq12 elements, and
    the signed-window Granger-Scott chain equals fields/mod.rs:35-46 on pairing values for edge and random scalars; the automatic
    choice equals the oracle for both kinds of input"""
    rng = np.random.default_rng(61)
    k = _fr(oracle, rng)
    g = oracle.pairing(oracle.g1_mul(oracle.g1_one(), k), oracle.g2_mul(oracle.g2_one(), _fr(oracle, rng)))
    rnd = _rf(oracle, rng, 12)
    U32 = hostsim_lib._U32P
    assert hs.lib.hsb_gt_is_cyclotomic(g.ctypes.data_as(U32)) == 1 and hs.lib.hsb_gt_is_cyclotomic(oracle.fq12_one().ctypes.data_as(U32)) == 1
    assert hs.lib.hsb_gt_is_cyclotomic(rnd.ctypes.data_as(U32)) == 0
    for kv in [0, 1, 2, 7, 8, 9, 15, 16, 0x88888888, M.R_ORD - 1, M.R_ORD - 2, (1 << 253) + 5] + [int.from_bytes(rng.bytes(40), "little") % M.R_ORD for _ in range(4)]:
        ke = oracle.fp_from_int(FR, kv)
        assert np.array_equal(hs.call("hsb_gt_pow_cyclotomic", g, ke, out_words=96), oracle.gt_pow(g, ke)), kv
        # the default chain: Frobenius decomposition (exact on order-r elements, which is what the reference's Gt holds)
        assert np.array_equal(hs.call("hsb_gt_pow_gls", g, ke, out_words=96), oracle.gt_pow(g, ke)), kv
    one = oracle.fq12_one()
    assert np.array_equal(hs.call("hsb_gt_pow_gls", one, oracle.fp_from_int(FR, M.R_ORD - 3), out_words=96), one)
    ke = _fr(oracle, rng)
    assert np.array_equal(hs.call("hsb_gt_pow_auto", g, ke, out_words=96), oracle.gt_pow(g, ke))
    assert np.array_equal(hs.call("hsb_gt_pow_auto", rnd, ke, out_words=96), oracle.gt_pow(rnd, ke))


def test_g2_gls_scalar_mul(oracle, hs):
    """bn254_g2_mul_batch's chain: the 4-dimensional GLS decomposition k = k0 + k1 L + k2 L^2 + k3 L^3 (L = q mod r, the eigenvalue of
    the reference's mul_by_q on G2) as the device computes it, and the whole scalar multiplication in both Fq2 mappings, normalized,
    against groups/mod.rs:250-270 - edge scalars (0, 1, r-1, the eigenvalue and its powers), a point with z = 1 and infinity"""
    rng = np.random.default_rng(62)
    lam = M.Q % M.R_ORD
    assert (pow(lam, 4, M.R_ORD) - pow(lam, 2, M.R_ORD) + 1) % M.R_ORD == 0
    b2 = oracle.g2_mul(oracle.g2_one(), oracle.fp_from_int(FR, 999))
    ks = [0, 1, 2, 3, 15, 16, 17, M.R_ORD - 1, M.R_ORD - 2, lam, lam + 1, M.R_ORD - lam, lam * lam % M.R_ORD, pow(lam, 3, M.R_ORD), 1 << 200, (1 << 253) + 7]
    ks += [int.from_bytes(rng.bytes(40), "little") % M.R_ORD for _ in range(8)]
    for kv in ks:
        k = oracle.fp_from_int(FR, kv)
        d = hs.call("hs_gls_decompose", k, out_words=16).view(np.uint32)
        parts = [(int(d[4 * i]) | int(d[4 * i + 1]) << 32 | int(d[4 * i + 2]) << 64, int(d[4 * i + 3])) for i in range(4)]
        assert sum((-m if s else m) * pow(lam, i, M.R_ORD) for i, (m, s) in enumerate(parts)) % M.R_ORD == kv
        assert all(m < (1 << 67) for m, _ in parts)
        for base in (b2, oracle.g2_one(), oracle.g2_zero()):
            want = canon_infinity(oracle.g2_normalize(oracle.g2_mul(base, k)))
            assert np.array_equal(hs.call("hs_g2_mul_gls", base, k, out_words=48), want), kv
            assert np.array_equal(hs.call("hsb_g2_mul_gls", base, k, out_words=48), want), kv


def test_shared_accumulator_miller_loop(oracle, hs):
    """the multi-pairing's Miller loop with ONE accumulator for m pairs (pairing.hpp miller_loop_shared: one f^2 per doubling step for
    all of them), bounds enforced: FE(shared value) == the oracle's fold of shootout/main.rs:11-16, also with an infinite pair"""
    rng = np.random.default_rng(63)
    for m in (2, 4):
        P = np.stack([oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng)) for _ in range(m)])
        Q = np.stack([oracle.g2_mul(oracle.g2_one(), _fr(oracle, rng)) for _ in range(m)])
        assert np.array_equal(hs.call("hsb_pairing_product_shared", m, P, Q, out_words=96), oracle.pairing_product(P, Q))
        P[m - 1] = oracle.g1_zero(); Q[0] = oracle.g2_zero()
        assert np.array_equal(hs.call("hsb_pairing_product_shared", m, P, Q, out_words=96), oracle.pairing_product(P, Q))


def test_quad_mapping_on_simulated_quad(oracle, hs, kats):
    """bn_amd/csrc/quad.hpp (one pairing on FOUR lanes: lower lane pair c0, upper c1) on the 4-lane value type of
    tests/hostsim/lanequad.hpp, every limb / value bound enforced: each split operation against the oracle or its lane-pair twin,
    the final exponentiation, and whole pairings - random, the reference's known answer (groups/mod.rs:773-796), infinity"""
    rng = np.random.default_rng(405)
    def fq12(): return np.concatenate([oracle.fp_from_int(FQ, int.from_bytes(rng.bytes(40), "little") % M.Q) for _ in range(12)])
    def fq2(): return np.concatenate([oracle.fp_from_int(FQ, int.from_bytes(rng.bytes(40), "little") % M.Q) for _ in range(2)])
    for _ in range(3):
        a, b = fq12(), fq12()
        assert np.array_equal(hs.call("hsq_fq12_sqr", a, out_words=96), oracle.fq12_sqr(a))                      # fq12.rs:275-282
        assert np.array_equal(hs.call("hsq_fq12_mul", a, b, 0, out_words=96), oracle.fq12_mul(a, b))             # fq12.rs:295-307
        bc = hs.call("hsq_fq12_conj", b, out_words=96)
        assert np.array_equal(hs.call("hsq_fq12_mul", a, b, 1, out_words=96), oracle.fq12_mul(a, bc))            # by the conjugate
        assert np.array_equal(hs.call("hsq_fq12_inverse", a, out_words=96), oracle.fq12_inverse(a))              # fq12.rs:284-292
        for pw in (1, 2, 3):
            assert np.array_equal(hs.call("hsq_fq12_frobenius", a, pw, out_words=96), hs.call("hsb_fq12_frobenius", a, pw, out_words=96))
        l0, lvw, lvv = fq2(), fq2(), fq2()
        assert np.array_equal(hs.call("hsq_fq12_mul_by_024", a, l0, lvw, lvv, out_words=96), hs.call("hsb_fq12_mul_by_024", a, l0, lvw, lvv, out_words=96))
    for v, z in ((oracle.fq12_one(), None), (np.zeros(48, np.uint64), None)):
        assert np.array_equal(hs.call("hsq_fq12_sqr", v, out_words=96), oracle.fq12_sqr(v))
    P = oracle.g1_mul(oracle.g1_one(), _fr(oracle, rng)); Q = oracle.g2_mul(oracle.g2_one(), _fr(oracle, rng))
    g = oracle.pairing(P, Q)
    assert np.array_equal(hs.call("hsq_fq12_cyclotomic_sqr", g, out_words=96), oracle.fq12_sqr(g))               # fq12.rs:178-227 on a Gt value
    assert np.array_equal(hs.call("hsq_final_exponentiation", oracle.miller_only(P, Q), out_words=96), g)        # fq12.rs:41-88
    assert np.array_equal(hs.call("hsq_pairing", P, Q, 1, out_words=96), g)
    # the Miller value of the NAF schedule differs from the reference's, its exponentiation does not
    assert np.array_equal(oracle.fq12_final_exponentiation(hs.call("hsq_pairing", P, Q, 0, out_words=96)), g)
    k = kats["test_reduced_pairing"]
    Pk = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_decimal(FR, k["k1"])); Qk = oracle.g2_mul(oracle.g2_one(), oracle.fp_from_decimal(FR, k["k2"]))
    assert oracle.fq12_to_ints(hs.call("hsq_pairing", Pk, Qk, 1, out_words=96)) == [int(x) for x in k["expected"]]
    one = oracle.fq12_one()
    assert np.array_equal(hs.call("hsq_pairing", oracle.g1_zero(), Q, 1, out_words=96), one)
    assert np.array_equal(hs.call("hsq_pairing", P, oracle.g2_zero(), 1, out_words=96), one)
    assert np.array_equal(hs.call("hsq_pairing", oracle.g1_one(), oracle.g2_one(), 1, out_words=96), oracle.pairing(oracle.g1_one(), oracle.g2_one()))

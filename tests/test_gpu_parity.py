"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI versus the CPU oracle, bit for bit."""
import numpy as np
import pytest

import bn_model as M
from bn_oracle import FQ, FR
from conftest import canon_infinity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import bn_amd
    return bn_amd.Engine(0)


def _scalars(rng, n):
    return [int.from_bytes(rng.bytes(64), "little") % M.R_ORD for _ in range(n)]


def _fr(oracle, vals):
    return np.stack([oracle.fp_from_int(FR, v) for v in vals])


def _points(oracle, rng, n):
    """Jacobian z != 1 points r*G1, s*G2 exactly as benches/api.rs builds inputs (G::random = one * Fr::random)"""
    k1 = _fr(oracle, _scalars(rng, n)); k2 = _fr(oracle, _scalars(rng, n))
    P = oracle.g1_mul_batch_jacobian(np.tile(oracle.g1_one(), (n, 1)), k1)
    Q = oracle.g2_mul_batch_jacobian(np.tile(oracle.g2_one(), (n, 1)), k2)
    return P, Q


def test_reference_kats_on_gpu(oracle, kats, eng):
    """groups/mod.rs:773-796 (test_reduced_pairing) through the GPU"""
    k1 = oracle.fp_from_decimal(FR, kats["test_reduced_pairing"]["k1"]); k2 = oracle.fp_from_decimal(FR, kats["test_reduced_pairing"]["k2"])
    P = oracle.g1_mul(oracle.g1_one(), k1); Q = oracle.g2_mul(oracle.g2_one(), k2)
    gt = eng.pairing_batch(P, Q)[0]
    assert oracle.fq12_to_ints(gt) == [int(x) for x in kats["test_reduced_pairing"]["expected"]]


def test_reference_miller_loop_kat_on_gpu(oracle, kats, eng):
    """groups/mod.rs:522-547 (test_miller_loop) literally: miller_loop(precompute(k2 G2), k1 G1) through bn254_miller_batch_dev - the
    reference-schedule kernel, whose un-exponentiated value must be the reference's twelve field elements - and through the one-lane test double"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    k = kats["test_miller_loop"]
    P = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_decimal(FR, k["k1"])); Q = oracle.g2_mul(oracle.g2_one(), oracle.fp_from_decimal(FR, k["k2"]))
    from testdouble import OneLaneEngine
    for mapping in (1, 0):                       # 1: the product's lane-pair kernels; 0: the one-lane test double (tests/testdouble/)
        e = bn_amd.Engine(0) if mapping == 1 else OneLaneEngine(0)
        te = D.TorchEngine(e, torch.device("cuda", 0))
        dp = torch.from_numpy(P.reshape(1, 12).view(np.int64)).to(te.device); dq = torch.from_numpy(Q.reshape(1, 24).view(np.int64)).to(te.device)
        f = te.empty(1, 48)
        e.miller_batch_dev(dp.data_ptr(), dq.data_ptr(), f.data_ptr(), 1, te._stream())
        torch.cuda.synchronize()
        assert oracle.fq12_to_ints(f.cpu().numpy().view(np.uint64)[0]) == [int(x) for x in k["expected"]], mapping
        # and the second half of the reference's test chain: final_exponentiation(that value) = test_reduced_pairing's answer
        e.final_exp_batch_dev(f.data_ptr(), f.data_ptr(), 1, te._stream())
        torch.cuda.synchronize()
        assert oracle.fq12_to_ints(f.cpu().numpy().view(np.uint64)[0]) == [int(x) for x in kats["test_reduced_pairing"]["expected"]], mapping
        e.close()


def test_reference_fq12_vector_on_gpu(oracle, kats, eng):
    """fields/mod.rs:83-169 (fq12_test_vector): every one of its 111 Fq12 products (100 x `next * start`, 10 + 1 x `squared`) runs on the
    device through bn254_gt_mul_batch (the element is an arbitrary Fq12, not a Gt value - the kernel is the general product); the ten
    rounds of add / sub / neg between them are not operations of the C ABI (Gt has no addition) and stay on the host"""
    k = kats["fq12_test_vector"]
    start = oracle.fq12_from_ints(k["start"]).reshape(1, 48)
    nxt = start.copy()
    for _ in range(100):
        nxt = eng.gt_mul_batch(nxt, start)
    cpy = nxt.copy()
    for _ in range(10):
        nxt = eng.gt_mul_batch(nxt, nxt)
    for _ in range(10):
        nxt = oracle.fq12_neg(oracle.fq12_sub(oracle.fq12_add(nxt[0], start[0]), cpy[0])).reshape(1, 48)
    nxt = eng.gt_mul_batch(nxt, nxt)
    assert oracle.fq12_to_ints(nxt[0]) == [int(x) for x in k["finally"]]
    # the same 100-product prefix as ONE batch of 64 copies: every lane pair of a wave agrees
    b = np.tile(start, (64, 1)); acc = b.copy()
    for _ in range(100):
        acc = eng.gt_mul_batch(acc, b)
    assert (acc == cpy).all()


def test_fq12_products_on_edge_coefficients(oracle, eng):
    """the general Fq12 product of the lane-pair kernels on elements whose twelve coefficients are drawn from {0, 1, 2, q-1, q-2, (q-1)/2,
    2^253, 2^29 - 1, 2^232, R mod q} and random values: the Karatsuba cross products run on SIGNED differences a_i - a_j (tower.hpp
    f2_cross, fe_mul2s) - equal coefficients make them zero, 0 against q-1 drives them to either extreme; squares (a * a) and products by
    one and by zero included.  Against oracle.fq12_mul, element for element"""
    rng = np.random.default_rng(77)
    edge = [0, 1, 2, M.Q - 1, M.Q - 2, (M.Q - 1) // 2, 1 << 253, (1 << 29) - 1, 1 << 232, M.MONT_R % M.Q]
    pool = edge + [int.from_bytes(rng.bytes(40), "little") % M.Q for _ in range(6)]
    n = 192
    def draw():
        return np.stack([oracle.fq12_from_ints([pool[i] for i in rng.integers(0, len(pool), 12)]) for _ in range(n)])
    a, b = draw(), draw()
    a[0] = oracle.fq12_from_ints([0] * 12); b[1] = oracle.fq12_from_ints([1] + [0] * 11); a[2] = oracle.fq12_from_ints([M.Q - 1] * 12); b[2] = oracle.fq12_from_ints([0, M.Q - 1] * 6)
    b[3] = a[3]; b[4] = a[4]                                                      # squares through the product
    want = np.stack([oracle.fq12_mul(x, y) for x, y in zip(a, b)])
    assert np.array_equal(eng.gt_mul_batch(a, b), want)
    assert np.array_equal(eng.gt_mul_batch(b, a), want)


def test_reference_cyclotomic_exp_kat_on_gpu(oracle, kats):
    """fields/mod.rs:171-201 (test_cyclotomic_exp): orig.exp_by_neg_z() on the device, the reference's operation sequence
    (bn254_exp_by_neg_z_dev) - the vector is OFF the cyclotomic subgroup, so only that sequence reproduces it"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    k = kats["test_cyclotomic_exp"]
    te = D.TorchEngine(bn_amd.Engine(0), torch.device("cuda", 0))
    n = 70                                                          # more than one wave of lane pairs, ragged
    a = torch.from_numpy(np.tile(oracle.fq12_from_ints(k["orig"]), (n, 1)).view(np.int64)).to(te.device)
    out = te.empty(n, 48)
    te.e.exp_by_neg_z_dev(a.data_ptr(), out.data_ptr(), n, te._stream())
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    assert oracle.fq12_to_ints(got[0]) == [int(x) for x in k["expected"]]
    assert (got == got[0]).all()
    # on a cyclotomic element the engine's own chain (inside the final exponentiation) and this one agree: f^(-u) both ways
    rng = np.random.default_rng(77)
    P, Q = _points(oracle, rng, 3)
    g = te.e.pairing_batch(P, Q)
    gd = torch.from_numpy(g.view(np.int64)).to(te.device); o2 = te.empty(3, 48)
    te.e.exp_by_neg_z_dev(gd.data_ptr(), o2.data_ptr(), 3, te._stream())
    torch.cuda.synchronize()
    want = np.stack([oracle.fq12_exp_by_neg_z(x) for x in g])
    assert np.array_equal(o2.cpu().numpy().view(np.uint64), want)


def test_options_api(eng):
    """bn254_ctx_set_option / get_option: defaults derive from the CU count, values are validated, negative restores the default"""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert eng.get_option("wave_pairing_max") == 14 * cus and eng.get_option("wave_fe_max") == 13 * cus
    assert eng.get_option("round_pairs") == 256 * cus and eng.get_option("pipeline_slots") == 2
    assert eng.get_option("miller_shared") == 0 and eng.get_option("gt_pow_mode") == 0
    eng.set_option("wave_pairing_max", 7); assert eng.get_option("wave_pairing_max") == 7
    eng.set_option("wave_pairing_max", -5); assert eng.get_option("wave_pairing_max") == 14 * cus
    for name, bad in (("miller_shared", 3), ("gt_pow_mode", 9), ("product_per_wave", 33), ("pipeline_slots", 5), ("product_bfly", 6)):
        with pytest.raises(Exception):
            eng.set_option(name, bad)
    with eng.options(product_chunk=3, pipeline_slots=1):
        assert eng.get_option("product_chunk") == 3 and eng.get_option("pipeline_slots") == 1
    assert eng.get_option("product_chunk") == -1 and eng.get_option("pipeline_slots") == 2
    # size options are capped at what one launch addresses (2^22); a rejected value inside options() leaves the context as it was
    for name in ("quad_max", "round_pairs", "wave_pairing_max", "wave_fe_max", "pipeline_chunk"):
        eng.set_option(name, 1 << 22)
        with pytest.raises(Exception):
            eng.set_option(name, (1 << 22) + 1)
        eng.set_option(name, None)
    eng.set_option("wave_fe_max", 99)
    with pytest.raises(Exception):
        with eng.options(wave_fe_max=5, quad_max=7, miller_shared=3):      # the third value is invalid
            pass
    assert eng.get_option("wave_fe_max") == 99 and eng.get_option("quad_max") == 64 * cus and eng.get_option("miller_shared") == 0
    eng.set_option("wave_fe_max", None)
    # bn254_ctx_set_mapping is kept for ABI compatibility: 1 is the only mapping, the one-lane test double (0) left the library in round 5
    assert eng._lib.bn254_ctx_set_mapping(eng._h, 1) == 0 and eng._lib.bn254_ctx_set_mapping(eng._h, 0) == -2


def test_pairing_batch_matches_oracle(oracle, eng):
    rng = np.random.default_rng(101)
    n = 200                                   # ragged: not a multiple of the wave size
    P, Q = _points(oracle, rng, n)
    got = eng.pairing_batch(P, Q)
    want = oracle.pairing_batch(P, Q)
    assert got.shape == (n, 48)
    assert np.array_equal(got, want)


def test_pairing_edge_cases(oracle, eng):
    rng = np.random.default_rng(102)
    P, Q = _points(oracle, rng, 6)
    P[1] = oracle.g1_zero(); Q[2] = oracle.g2_zero(); P[3] = oracle.g1_zero(); Q[3] = oracle.g2_zero()   # infinity -> one
    P[4] = oracle.g1_one(); Q[4] = oracle.g2_one()                                                      # z == 1 shortcut inputs
    P[5] = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_int(FR, M.R_ORD - 1))
    got = eng.pairing_batch(P, Q)
    assert np.array_equal(got, oracle.pairing_batch(P, Q))
    one = oracle.fq12_one()
    for i in (1, 2, 3):
        assert np.array_equal(got[i], one)
    assert eng.pairing_batch(np.zeros((0, 12), np.uint64), np.zeros((0, 24), np.uint64)).shape == (0, 48)   # empty batch


def test_scalar_mul_matches_oracle(oracle, eng):
    rng = np.random.default_rng(103)
    n = 70
    ks = _scalars(rng, n); ks[:6] = [0, 1, 2, M.R_ORD - 1, M.R_ORD - 2, 1 << 200]
    k = _fr(oracle, ks)
    P, Q = _points(oracle, rng, n)
    P[7] = oracle.g1_zero(); Q[7] = oracle.g2_zero(); P[8] = oracle.g1_one(); Q[8] = oracle.g2_one()
    assert np.array_equal(eng.g1_mul_batch(P, k), canon_infinity(oracle.g1_mul_batch(P, k)))
    assert np.array_equal(eng.g2_mul_batch(Q, k), canon_infinity(oracle.g2_mul_batch(Q, k)))


def test_pairing_product_matches_fold(oracle, eng):
    rng = np.random.default_rng(104)
    for n in (0, 1, 5, 130):
        P, Q = _points(oracle, rng, n) if n else (np.zeros((0, 12), np.uint64), np.zeros((0, 24), np.uint64))
        if n > 3:
            P[2] = oracle.g1_zero()             # a pair with a point at infinity contributes one
        got = eng.pairing_product(P, Q)
        assert np.array_equal(got, oracle.pairing_product(P, Q)), n


def test_bilinearity_on_gpu(oracle, eng):
    """groups/mod.rs:798-823 (test_binlinearity): e(sP,Q) == e(P,sQ) == e(P,Q)^s, != 1"""
    rng = np.random.default_rng(105)
    n = 8
    P, Q = _points(oracle, rng, n)
    s = _fr(oracle, _scalars(rng, n))
    sP = eng.g1_mul_batch(P, s); sQ = eng.g2_mul_batch(Q, s)
    a = eng.pairing_batch(sP, Q); b = eng.pairing_batch(P, sQ); c = eng.pairing_batch(P, Q)
    assert np.array_equal(a, b)
    one = oracle.fq12_one()
    for i in range(n):
        assert np.array_equal(oracle.gt_pow(c[i], s[i]), a[i])
        assert not np.array_equal(a[i], one)


def test_reference_api_mirror(oracle):
    """examples/joux.rs through the mirrored API (bn_amd.api)"""
    import bn_amd
    rng = np.random.default_rng(106)
    a, b, c = (bn_amd.Fr.random(rng) for _ in range(3))
    pa, qa = bn_amd.G1.one() * a, bn_amd.G2.one() * a
    pb, qb = bn_amd.G1.one() * b, bn_amd.G2.one() * b
    pc, qc = bn_amd.G1.one() * c, bn_amd.G2.one() * c
    # each party's shared key e(P_b, Q_c)^a etc. - compare through the oracle's Gt::pow
    ka = oracle.gt_pow(bn_amd.pairing(pb, qc).limbs, a.limbs)
    kb = oracle.gt_pow(bn_amd.pairing(pc, qa).limbs, b.limbs)
    kc = oracle.gt_pow(bn_amd.pairing(pa, qb).limbs, c.limbs)
    assert np.array_equal(ka, kb) and np.array_equal(kb, kc)
    assert bn_amd.pairing(bn_amd.G1.zero(), bn_amd.G2.one()) == bn_amd.Gt.one()
    assert (bn_amd.G1.one() * bn_amd.Fr(5)) == (bn_amd.G1.one() * bn_amd.Fr(2)) * bn_amd.Fr(3) * bn_amd.Fr(5) * bn_amd.Fr(6).inverse()


@pytest.mark.parametrize("mapping", [0, 1])
def test_both_lane_mappings_agree_with_oracle(oracle, mapping):
    """mapping 1: the product (one lane PAIR per pairing, Fq2B, DPP exchange); mapping 0: the one-lane test double of tests/testdouble/
    (Fq2A: the same tower / pairing / curve templates over the other Fq2) - same bytes"""
    import bn_amd
    from testdouble import OneLaneEngine
    e = bn_amd.Engine(0) if mapping == 1 else OneLaneEngine(0)
    rng = np.random.default_rng(107)
    n = 97                                    # odd: the last wave has an unpaired tail
    P, Q = _points(oracle, rng, n)
    P[5] = oracle.g1_zero(); Q[6] = oracle.g2_zero(); P[7] = oracle.g1_one(); Q[7] = oracle.g2_one()
    assert np.array_equal(e.pairing_batch(P, Q), oracle.pairing_batch(P, Q))
    assert np.array_equal(e.pairing_product(P[:33], Q[:33]), oracle.pairing_product(P[:33], Q[:33]))
    # G * Fr: the test double = call-based kernels with G2 over Fq2A and plain 4-bit windows, the product = inlined GLV / GLS kernels with G2 on lane pairs
    k = _fr(oracle, _scalars(rng, n))
    assert np.array_equal(e.g1_mul_batch(P, k), canon_infinity(oracle.g1_mul_batch(P, k)))
    assert np.array_equal(e.g2_mul_batch(Q, k), canon_infinity(oracle.g2_mul_batch(Q, k)))


def test_device_resident_path_and_input_generator(oracle):
    """what bench.py runs: inputs generated on the device by the reference's double-and-add chain (bit-identical Jacobian
    limbs to the oracle's), pairings through the device-pointer API on torch tensors, and the sharded product on one rank"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    eng = D.TorchEngine(bn_amd.Engine(0), dev)
    lo, hi = 1000, 1000 + 150
    P, Q = D.synthetic_points(eng, lo, hi)
    k1 = D.synthetic_scalars(lo, hi, 0); k2 = D.synthetic_scalars(lo, hi, 1)
    n = hi - lo
    Pn = P.cpu().numpy().view(np.uint64); Qn = Q.cpu().numpy().view(np.uint64)
    assert np.array_equal(Pn, oracle.g1_mul_batch_jacobian(np.tile(oracle.g1_one(), (n, 1)), k1))
    assert np.array_equal(Qn, oracle.g2_mul_batch_jacobian(np.tile(oracle.g2_one(), (n, 1)), k2))
    out = D.pairing_batch_sharded(eng, P, Q)
    torch.cuda.synchronize()
    want = oracle.pairing_batch(Pn, Qn)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
    gt = D.pairing_product_sharded(eng, P, Q)
    torch.cuda.synchronize()
    assert np.array_equal(gt.cpu().numpy().view(np.uint64), oracle.pairing_product(Pn, Qn))


def test_full_size_batch_properties(oracle):
    """BASELINE.json configs[1] size (2^16): the whole batch against the oracle, both lane mappings agree on the
    whole batch, determinism, and product(batch) == pairing_product (size-independent checks)"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    n = 1 << 16
    from testdouble import OneLaneEngine
    engB = D.TorchEngine(bn_amd.Engine(0), dev)
    engA = D.TorchEngine(OneLaneEngine(0), dev)                 # the one-lane test double: a second implementation at full size
    P, Q = D.synthetic_points(engB, 0, n)
    outB = D.pairing_batch_sharded(engB, P, Q); outB2 = D.pairing_batch_sharded(engB, P, Q); outA = D.pairing_batch_sharded(engA, P, Q)
    torch.cuda.synchronize()
    assert torch.equal(outB, outB2) and torch.equal(outA, outB)
    # SURVEY 8d parity protocol: memcmp of the FULL 2^16 batch against the multi-threaded CPU oracle (~15 s on 16 host threads);
    # on a host with few cores a 2048-index random sample instead
    import bn_oracle
    cores = bn_oracle.usable_cpus()
    idx = np.arange(n) if cores >= 8 else np.random.default_rng(5).choice(n, 2048, replace=False)
    Pn = P.cpu().numpy().view(np.uint64)[idx]; Qn = Q.cpu().numpy().view(np.uint64)[idx]
    assert np.array_equal(outB.cpu().numpy().view(np.uint64)[idx], oracle.pairing_batch(Pn, Qn))
    prod_of_batch = engB.gt_product(outB)                       # product of the 2^16 reduced pairings ...
    prod = D.pairing_product_sharded(engB, P, Q)                # ... equals the multi-pairing with ONE final exponentiation
    torch.cuda.synchronize()
    assert torch.equal(prod_of_batch, prod)


def test_g1_mul_full_size_config5(oracle):
    """BASELINE.json configs[4]: 2^20 G1 scalar multiplications of DISTINCT random points by DISTINCT random Fr on one GPU
    (benches/api.rs:107-111; windowed kernel, normalized output): a 16384-index sample against the oracle, and the two device
    algorithms (windowed vs the reference's own chain) agree on all 2^20 after normalization"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    te = D.TorchEngine(bn_amd.Engine(0), dev)
    n = 1 << 20
    g1, _ = D.generator_limbs()
    kb = D.synthetic_scalars_device(te, 0, n, 0)
    base = te.empty(n, 12)
    te.e.tile_dev(torch.from_numpy(g1.view(np.int64)).to(dev).data_ptr(), 96, n, base.data_ptr(), te._stream())
    P = te.g1_mul(base, kb, normalize=False)                          # 2^20 distinct random points, Jacobian z != 1
    k = D.synthetic_scalars_device(te, 1 << 24, (1 << 24) + n, 1)     # 2^20 distinct scalars
    out = te.g1_mul(P, k, normalize=True)
    jac = te.g1_mul(P, k, normalize=False)                            # reference chain, raw Jacobian
    one = te.empty(n, 4)
    te.e.tile_dev(torch.from_numpy(oracle.fp_from_int(FR, 1).view(np.int64)).to(dev).data_ptr(), 32, n, one.data_ptr(), te._stream())
    out2 = te.g1_mul(jac, one, normalize=True)                        # normalize(chain result) via * 1
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    kn_all = k.cpu().numpy().view(np.uint64)
    assert np.unique(kn_all, axis=0).shape[0] == n                    # really 2^20 distinct scalars
    idx = np.random.default_rng(9).choice(n, 16384, replace=False)
    Pn = P.cpu().numpy().view(np.uint64)[idx]; kn = kn_all[idx]
    assert np.array_equal(out.cpu().numpy().view(np.uint64)[idx], canon_infinity(oracle.g1_mul_batch(Pn, kn)))


def test_gt_mul_and_pow_match_oracle(oracle, eng):
    """Gt * Gt (lib.rs:175-179) and Gt::pow (lib.rs:171) on the GPU, then test_binlinearity (groups/mod.rs:798-823) entirely
    through the mirrored API: e(P,Q)^s == e(sP,Q) == e(P,sQ), a != 1, a^(-1) * a == 1"""
    import bn_amd
    rng = np.random.default_rng(108)
    n = 9
    P, Q = _points(oracle, rng, n)
    g = eng.pairing_batch(P, Q)
    sv = _scalars(rng, n); sv[:3] = [0, 1, M.R_ORD - 1]
    s = _fr(oracle, sv)
    pw = eng.gt_pow_batch(g, s)
    ml = eng.gt_mul_batch(g, g[::-1].copy())
    for i in range(n):
        assert np.array_equal(pw[i], oracle.gt_pow(g[i], s[i]))
        assert np.array_equal(ml[i], oracle.fq12_mul(g[i], g[n - 1 - i]))
    p, q, sc = bn_amd.G1.random(rng), bn_amd.G2.random(rng), bn_amd.Fr.random(rng)
    a = bn_amd.pairing(p, q).pow(sc)
    assert a == bn_amd.pairing(p * sc, q) == bn_amd.pairing(p, q * sc)
    assert a != bn_amd.Gt.one()
    assert a.pow(-bn_amd.Fr.one()) * a == bn_amd.Gt.one()


def test_prepared_g2_mode(oracle, kats, eng):
    """groups/mod.rs:637-762 (test_prepared_g2): all 102 line coefficients computed ON THE GPU equal the reference's known
    answers; then pairing through the prepared coefficients (shared and per-pair) equals pairing()"""
    k2 = oracle.fp_from_decimal(FR, kats["test_prepared_g2"]["k2"])
    Q = oracle.g2_mul(oracle.g2_one(), k2)
    co = eng.g2_precompute(Q)[0]
    assert co.shape == (102, 24)
    for got, exp in zip(co, kats["test_prepared_g2"]["coeffs"]):
        for slot, name in enumerate(("ell_0", "ell_vw", "ell_vv")):
            assert [oracle.fp_to_int(FQ, got[8 * slot + 4 * i:8 * slot + 4 * i + 4]) for i in range(2)] == [int(x) for x in exp[name]]
    rng = np.random.default_rng(109)
    n = 37
    P, Qs = _points(oracle, rng, n)
    P[4] = oracle.g1_zero()
    # shared Q
    got = eng.pairing_prepared_batch(P, co)
    want = oracle.pairing_batch(P, np.tile(Q, (n, 1)))
    assert np.array_equal(got, want)
    # one prepared point per pair; coefficients equal the oracle's precompute
    cos = eng.g2_precompute(Qs)
    for i in (0, n - 1):
        assert np.array_equal(cos[i].reshape(102, 3, 8), oracle.g2_precompute(oracle.g2_to_affine(Qs[i])))
    assert np.array_equal(eng.pairing_prepared_batch(P, cos), oracle.pairing_batch(P, Qs))


def test_wire_format_on_gpu(oracle, eng):
    """SURVEY 8f-3 on the GPU: batch encode/decode of G1/G2 records incl. every validation outcome vs the code-derived oracle"""
    from conftest import g2_point_outside_subgroup
    rng = np.random.default_rng(110)
    n = 70
    P, Q = _points(oracle, rng, n)
    P[3] = oracle.g1_zero(); Q[4] = oracle.g2_zero(); P[5] = oracle.g1_one(); Q[5] = oracle.g2_one()
    e1 = eng.g1_encode_batch(P); e2 = eng.g2_encode_batch(Q)
    for i in range(n):
        assert np.array_equal(e1[i], oracle.g1_encode(P[i])) and np.array_equal(e2[i], oracle.g2_encode(Q[i]))
    # corrupt some records: bad tag, coordinate >= q, >= q^2, off-curve, outside the subgroup
    b1 = e1.copy(); b2 = e2.copy()
    b1[10, 0] = 9; b1[11, 1:33] = 255; b1[12, 40] ^= 1
    b2[10, 0] = 1; b2[11, 1:65] = 255; b2[12, 100] ^= 1; b2[13] = g2_point_outside_subgroup()
    d1, s1 = eng.g1_decode_batch(b1); d2, s2 = eng.g2_decode_batch(b2)
    for i in range(n):
        rc, want = oracle.g1_decode(b1[i]); assert s1[i] == rc
        assert np.array_equal(d1[i], want if rc == 0 else oracle.g1_zero())
        rc, want = oracle.g2_decode(b2[i]); assert s2[i] == rc
        assert np.array_equal(d2[i], want if rc == 0 else oracle.g2_zero())
    assert list(s1[10:13]) == [3, 1, 4] and list(s2[10:14]) == [3, 2, 4, 5]
    # Fr records
    k = _fr(oracle, [0, 1, M.R_ORD - 1] + _scalars(rng, 60))
    ek = eng.fr_encode_batch(k)
    for i in range(k.shape[0]):
        assert np.array_equal(ek[i], oracle.fr_encode(k[i]))
    bk = ek.copy(); bk[5] = np.frombuffer(M.R_ORD.to_bytes(32, "big"), np.uint8); bk[6] = 255
    dk, sk = eng.fr_decode_batch(bk)
    assert list(sk[4:8]) == [0, 1, 1, 0] and not dk[5].any() and not dk[6].any()
    mask = np.ones(k.shape[0], bool); mask[5:7] = False
    assert np.array_equal(dk[mask], k[mask])
    assert np.array_equal(d1[7], oracle.g1_normalize(P[7])) and np.array_equal(d2[7], oracle.g2_normalize(Q[7]))


@pytest.mark.parametrize("mapping", [0, 1])
def test_miller_values_equal_reference_schedule(oracle, mapping):
    """bn254_miller_batch_dev keeps the reference's schedule (groups/mod.rs:486-519), so the un-exponentiated values equal the
    oracle's limb for limb; bn254_final_exp_batch_dev of them is the pairing (the pairing kernels use the NAF schedule)"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    from testdouble import OneLaneEngine
    e = bn_amd.Engine(0) if mapping == 1 else OneLaneEngine(0)
    rng = np.random.default_rng(77)
    n = 33
    P, Q = _points(oracle, rng, n)
    tp = torch.from_numpy(P.view(np.int64)).to(dev); tq = torch.from_numpy(Q.view(np.int64)).to(dev)
    f = torch.empty(n, 48, dtype=torch.int64, device=dev); g = torch.empty_like(f)
    e.miller_batch_dev(tp.data_ptr(), tq.data_ptr(), f.data_ptr(), n, torch.cuda.current_stream(dev).cuda_stream)
    e.final_exp_batch_dev(f.data_ptr(), g.data_ptr(), n, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    fn = f.cpu().numpy().view(np.uint64); gn = g.cpu().numpy().view(np.uint64)
    for i in range(n):
        assert np.array_equal(fn[i], oracle.miller_only(P[i], Q[i]))
    assert np.array_equal(gn, oracle.pairing_batch(P, Q))


def test_golden_fixtures_on_gpu(eng, goldens):
    """the committed fixtures (tests/golden/pairing_goldens.npz: edge scalars 1, 2, r-1, ... then seeded random) through every
    host-buffer entry point - this test does not touch the oracle at all"""
    import os
    g = goldens
    n = g["k1"].shape[0]
    assert np.array_equal(eng.pairing_batch(g["g1"], g["g2"]), g["gt"])                    # a batch this small: one pairing per wave
    with eng.options(wave_pairing_max=0, wave_fe_max=0, quad_max=0):
        assert np.array_equal(eng.pairing_batch(g["g1"], g["g2"]), g["gt"])                # and through the lane-pair kernels
    one1 = np.tile(g["g1"][0], (n, 1)); one2 = np.tile(g["g2"][10], (n, 1))          # scalars1[0] == scalars2[10] == 1: the generators
    assert int(g["scalars1"][0]) == 1 and int(g["scalars2"][10]) == 1
    assert np.array_equal(eng.g1_mul_batch(one1, g["k1"]), g["g1"]) and np.array_equal(eng.g2_mul_batch(one2, g["k2"]), g["g2"])
    assert np.array_equal(eng.g2_precompute(g["g2"][5:6]).reshape(102, 24), g["coeffs"])
    assert np.array_equal(eng.g1_encode_batch(g["g1"][:32]), g["wire_g1"]) and np.array_equal(eng.g2_encode_batch(g["g2"][:32]), g["wire_g2"])
    d1, s1 = eng.g1_decode_batch(g["wire_g1"]); d2, s2 = eng.g2_decode_batch(g["wire_g2"])
    assert not s1.any() and not s2.any() and np.array_equal(d1, g["g1"][:32]) and np.array_equal(d2, g["g2"][:32])
    # e(a G1, b G2) = e(G1, G2)^(ab) on the device: Gt::pow of golden (1, 1)... index with scalars (1, x): gt[0] = e(G1, s2[0] G2)
    prod = eng.pairing_product(g["g1"], g["g2"])
    acc = g["gt"][0:1]
    for i in range(1, n):
        acc = eng.gt_mul_batch(acc, g["gt"][i:i + 1])
    assert np.array_equal(prod.reshape(1, 48), acc)


def test_group_addition_matches_reference_limbs(oracle, eng):
    """`G + G`, `G - G`, `-G` (lib.rs:103-114,146-157): raw Jacobian limbs incl. the zero / equal-point branches"""
    import bn_amd
    rng = np.random.default_rng(111)
    n = 40
    A1, A2 = _points(oracle, rng, n); B1, B2 = _points(oracle, rng, n)
    B1[1] = A1[1]; B2[1] = A2[1]                                    # equal points -> doubling branch
    A1[2] = oracle.g1_zero(); A2[2] = oracle.g2_zero(); B1[3] = oracle.g1_zero(); B2[3] = oracle.g2_zero()
    A1[4] = oracle.g1_zero(); B1[4] = oracle.g1_zero(); A2[4] = oracle.g2_zero(); B2[4] = oracle.g2_zero()
    s1 = eng.g1_add_batch(A1, B1); d1 = eng.g1_add_batch(A1, B1, negate_b=True)
    s2 = eng.g2_add_batch(A2, B2); d2 = eng.g2_add_batch(A2, B2, negate_b=True)
    for i in range(n):
        assert np.array_equal(s1[i], oracle.g1_add(A1[i], B1[i])) and np.array_equal(d1[i], oracle.g1_add(A1[i], oracle.g1_neg(B1[i])))
        assert np.array_equal(s2[i], oracle.g2_add(A2[i], B2[i])) and np.array_equal(d2[i], oracle.g2_add(A2[i], oracle.g2_neg(B2[i])))
    p = bn_amd.G1(A1[0]); q = bn_amd.G1(B1[0])
    assert np.array_equal((-q).limbs, oracle.g1_neg(B1[0])) and (p + q) - q == p and (p - p).is_zero()


# ------------------------------------------------------------------------------------------------ round 2: sizes, threads, multi-device
def test_synthetic_scalars_device_equals_numpy(oracle):
    """bn254_synthetic_scalars_dev (inputs generated in HBM) == bn_amd.distributed.synthetic_scalars word for word"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    te = D.TorchEngine(bn_amd.Engine(0), dev)
    for lo, hi, which in ((0, 300, 0), (12345, 12345 + 257, 1), ((1 << 24) + 7, (1 << 24) + 71, 1)):
        got = D.synthetic_scalars_device(te, lo, hi, which)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy().view(np.uint64), D.synthetic_scalars(lo, hi, which))


def test_default_context_is_thread_safe(oracle):
    """lib.rs:55-61: `pairing` is a pure function, the types are Send + Sync.  6 host threads call the C ABI with ctx == NULL (the
    process-wide default context) at the same time, with different batch sizes, several times each"""
    import ctypes as C
    import threading
    from bn_amd import _native
    lib = _native.lib()
    rng = np.random.default_rng(201)
    sizes = [3, 70, 129, 33, 250, 64]
    data = []
    for n in sizes:
        P, Q = _points(oracle, rng, n)
        data.append((P, Q, oracle.pairing_batch(P, Q)))
    errs = []
    def work(i):
        P, Q, want = data[i]
        for _ in range(4):
            out = np.zeros((P.shape[0], 48), np.uint64)
            rc = lib.bn254_pairing_batch(None, C.c_void_p(P.ctypes.data), C.c_void_p(Q.ctypes.data), C.c_void_p(out.ctypes.data), P.shape[0])
            if rc != 0 or not np.array_equal(out, want):
                errs.append((i, rc))
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(sizes))]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs


def test_two_streams_on_one_context(oracle):
    """the *_dev entry points share context-owned scratch (final-exponentiation table, product workspace): calls on two streams of
    ONE context must serialise on it (event-ordered), not race"""
    import torch
    import bn_amd
    dev = torch.device("cuda", 0)
    e = bn_amd.Engine(0)
    rng = np.random.default_rng(202)
    n = 4096
    P, Q = _points(oracle, rng, 64)
    P = np.tile(P, (n // 64, 1)); Q = np.tile(Q, (n // 64, 1))
    P2 = np.ascontiguousarray(P[::-1]); Q2 = np.ascontiguousarray(Q[::-1])
    want = oracle.pairing_batch(P[:64], Q[:64])
    tp, tq, tp2, tq2 = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (P, Q, P2, Q2))
    o1 = torch.empty(n, 48, dtype=torch.int64, device=dev); o2 = torch.empty_like(o1)
    pr1 = torch.empty(48, dtype=torch.int64, device=dev); pr2 = torch.empty_like(pr1)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    for _ in range(3):
        e.pairing_batch_dev(tp.data_ptr(), tq.data_ptr(), o1.data_ptr(), n, s1.cuda_stream)
        e.pairing_batch_dev(tp2.data_ptr(), tq2.data_ptr(), o2.data_ptr(), n, s2.cuda_stream)
        e.miller_product_dev(tp.data_ptr(), tq.data_ptr(), n, pr1.data_ptr(), s1.cuda_stream)
        e.miller_product_dev(tp2.data_ptr(), tq2.data_ptr(), n, pr2.data_ptr(), s2.cuda_stream)
    torch.cuda.synchronize()
    a = o1.cpu().numpy().view(np.uint64); b = o2.cpu().numpy().view(np.uint64)
    assert np.array_equal(a[:64], want) and np.array_equal(a, np.tile(want, (n // 64, 1)))
    assert np.array_equal(b, a[::-1])
    assert torch.equal(pr1, pr2)                 # same multiset of pairs -> same product (Fq12 products commute, values canonical)


def test_host_buffer_pipeline_matches_device_path(oracle):
    """bn254_pairing_batch / g*_mul_batch on pageable host buffers run chunked over several streams with pinned staging: a ragged
    multi-chunk batch equals the single-launch device path everywhere and the oracle on a sample"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    e = bn_amd.Engine(0)
    te = D.TorchEngine(e, dev)
    n = 20000 + 37
    P, Q = D.synthetic_points(te, 5000, 5000 + n)
    ref = te.pairing_batch(P, Q)
    torch.cuda.synchronize()
    Pn = P.cpu().numpy().view(np.uint64); Qn = Q.cpu().numpy().view(np.uint64)
    got = e.pairing_batch(Pn, Qn)
    assert np.array_equal(got, ref.cpu().numpy().view(np.uint64))
    idx = np.random.default_rng(7).choice(n, 256, replace=False)
    assert np.array_equal(got[idx], oracle.pairing_batch(Pn[idx], Qn[idx]))
    # just above one chunk: two equal chunks on the two streams (no ragged one-wave tail launch); result buffer supplied
    n2 = 65536 + 40
    P2, Q2 = D.synthetic_points(te, 70000, 70000 + n2)
    ref2 = te.pairing_batch(P2, Q2)
    torch.cuda.synchronize()
    out2 = np.zeros((n2, 48), np.uint64)
    assert e.pairing_batch(P2.cpu().numpy().view(np.uint64), Q2.cpu().numpy().view(np.uint64), out2) is out2
    assert np.array_equal(out2, ref2.cpu().numpy().view(np.uint64))
    k = D.synthetic_scalars(100, 100 + n, 1)
    m1 = e.g1_mul_batch(Pn, k)
    assert np.array_equal(m1, te.g1_mul(P, torch.from_numpy(k.view(np.int64)).to(dev)).cpu().numpy().view(np.uint64))
    assert np.array_equal(m1[idx], canon_infinity(oracle.g1_mul_batch(Pn[idx], k[idx])))


def test_engine_argument_checks(oracle):
    """binary batch methods must reject operands of different length instead of reading past the shorter buffer; a
    closed Engine must raise instead of falling through to the C ABI's NULL = default context"""
    import bn_amd
    from bn_amd import _native
    e = bn_amd.Engine(0)
    rng = np.random.default_rng(203)
    P, Q = _points(oracle, rng, 4)
    k = _fr(oracle, _scalars(rng, 3))
    g = e.pairing_batch(P, Q)
    for fn, a, b in ((e.g1_mul_batch, P, k), (e.g2_mul_batch, Q, k), (e.g1_add_batch, P, P[:3]), (e.g2_add_batch, Q, Q[:2]),
                     (e.gt_mul_batch, g, g[:1]), (e.gt_pow_batch, g, k), (e.pairing_batch, P, Q[:3])):
        with pytest.raises(ValueError):
            fn(a, b)
    e.close()
    with pytest.raises(_native.Bn254Error):
        e.pairing_batch(P, Q)


def test_gt_inverse_and_windowed_pow(oracle, eng):
    """Gt::inverse (lib.rs:172) and the windowed Gt::pow on edge exponents (0, 1, 15, 16, r-1, 2^252..) vs the oracle's bit-serial pow"""
    rng = np.random.default_rng(204)
    n = 40
    P, Q = _points(oracle, rng, n)
    g = eng.pairing_batch(P, Q)
    sv = _scalars(rng, n)
    sv[:10] = [0, 1, 2, 15, 16, 17, M.R_ORD - 1, 1 << 252, (1 << 253) - 1, 0xf0f0f0f0f0f0f0f0f0f0]
    s = _fr(oracle, sv)
    pw = eng.gt_pow_batch(g, s)
    inv = eng.gt_inverse_batch(g)
    one = oracle.fq12_one()
    for i in range(n):
        assert np.array_equal(pw[i], oracle.gt_pow(g[i], s[i])), i
        assert np.array_equal(inv[i], oracle.fq12_inverse(g[i]))
    assert np.array_equal(eng.gt_mul_batch(g, inv), np.tile(one, (n, 1)))
    # an element OUTSIDE the cyclotomic subgroup (a raw Miller value): the windowed chain uses general squarings, so still exact
    raw = np.stack([oracle.miller_only(P[i], Q[i]) for i in range(4)])
    assert np.array_equal(eng.gt_pow_batch(raw, s[10:14]), np.stack([oracle.gt_pow(raw[i], s[10 + i]) for i in range(4)]))
    import bn_amd
    a = bn_amd.Gt(g[0])
    assert a.inverse() * a == bn_amd.Gt.one() and a.inverse() == a.pow(-bn_amd.Fr.one())


def test_config3_shard_size_2_17(oracle):
    """BASELINE.json configs[2]: 2^20 pairings over 8 GPUs = 2^17 per GPU.  The per-GPU shard on one GPU: a 4096-index sample
    against the oracle, determinism, both lane mappings agree on all 2^17, and the shard equals the same indices of two 2^16 halves"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    n = 1 << 17
    lo = 3 * n                                   # the shard GPU 3 of 8 would own
    from testdouble import OneLaneEngine
    engB = D.TorchEngine(bn_amd.Engine(0), dev)
    engA = D.TorchEngine(OneLaneEngine(0), dev)                 # the one-lane test double: a second implementation at full size
    P, Q = D.synthetic_points(engB, lo, lo + n)
    outB = engB.pairing_batch(P, Q); outB2 = engB.pairing_batch(P, Q); outA = engA.pairing_batch(P, Q)
    h0 = engB.pairing_batch(P[:n // 2].contiguous(), Q[:n // 2].contiguous()); h1 = engB.pairing_batch(P[n // 2:].contiguous(), Q[n // 2:].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(outB, outB2) and torch.equal(outA, outB) and torch.equal(torch.cat([h0, h1]), outB)
    idx = np.sort(np.random.default_rng(11).choice(n, 4096, replace=False))
    Pn = P.cpu().numpy().view(np.uint64)[idx]; Qn = Q.cpu().numpy().view(np.uint64)[idx]
    assert np.array_equal(outB.cpu().numpy().view(np.uint64)[idx], oracle.pairing_batch(Pn, Qn))


def test_config4_product_sizes(oracle):
    """BASELINE.json configs[3]: multi-pairing product of 2^18 pairs (2^15 per GPU on 8).  On one GPU: the 2^18-pair product (ONE
    final exponentiation) equals the Fq12 product of the 2^18 reduced pairings; the 2^15 shard likewise, and the fold of a
    512-pair sample equals the oracle's fold of shootout/main.rs:11-16; the 8-way sharded combination (8 partials -> gt_product
    -> one final exponentiation, what the RCCL path computes after its all-gather) equals the single-GPU product"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    te = D.TorchEngine(bn_amd.Engine(0), dev)
    n = 1 << 18
    P, Q = D.synthetic_points(te, 0, n)
    prod = D.pairing_product_sharded(te, P, Q)                       # world 1: miller_product -> final_exp
    batch = te.pairing_batch(P, Q)
    assert torch.equal(prod, te.gt_product(batch))
    sh = n // 8
    parts = torch.stack([te.miller_product(P[g * sh:(g + 1) * sh].contiguous(), Q[g * sh:(g + 1) * sh].contiguous()) for g in range(8)])
    assert torch.equal(te.final_exp(te.gt_product(parts)), prod)
    p15 = D.pairing_product_sharded(te, P[:sh].contiguous(), Q[:sh].contiguous())
    assert torch.equal(p15, te.gt_product(batch[:sh].contiguous()))
    torch.cuda.synchronize()
    Pn = P[:512].cpu().numpy().view(np.uint64); Qn = Q[:512].cpu().numpy().view(np.uint64)
    got = D.pairing_product_sharded(te, P[:512].contiguous(), Q[:512].contiguous())
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint64), oracle.pairing_product(Pn, Qn))


def test_multi_device_c_abi(oracle):
    """include/bn254_hip.h bn254_multi_*: the N > 1 split a Rust/C host gets without Python.  Two ranks on ONE GPU (device list
    [0, 0]: contexts, host threads, shards, the gather and the single final exponentiation are all real; the exchange is peer
    copies because RCCL needs one GPU per rank), and one rank through RCCL itself (ncclCommInitAll + ncclAllGather of 48 x u64)."""
    import bn_amd
    rng = np.random.default_rng(205)
    n = 133
    P, Q = _points(oracle, rng, n)
    P[5] = oracle.g1_zero()
    want_b = oracle.pairing_batch(P, Q); want_p = oracle.pairing_product(P, Q)
    for devs, kind in (([0, 0], "peer"), ([0, 0, 0], "peer"), ([0], None)):
        m = bn_amd.MultiEngine(devs)
        if kind:
            assert m.exchange == kind
        assert np.array_equal(m.pairing_batch(P, Q), want_b), devs
        assert np.array_equal(m.pairing_product(P, Q), want_p), devs
        assert np.array_equal(m.pairing_product(P[:1], Q[:1]), oracle.pairing_product(P[:1], Q[:1]))     # fewer pairs than ranks
        assert np.array_equal(m.pairing_product(P[:0], Q[:0]), oracle.fq12_one())
        if devs == [0]:
            print("exchange with one rank:", m.exchange)
            assert m.exchange == "rccl", "RCCL did not load / ncclCommInitAll failed on one device"
        m.close()


GPU_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [%(root)r, %(root)r + "/oracle", %(root)r + "/tests"]
import bn_amd
from bn_amd import distributed as D
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
eng = D.TorchEngine(bn_amd.Engine(0), dev)                       # the real HIP engine, both ranks on the one GPU
n = %(n)d
lo, hi = D.shard_range(n, rank, world)
P, Q = D.synthetic_points(eng, lo, hi)
gt = D.pairing_product_sharded(eng, P, Q)
loc = D.pairing_batch_sharded(eng, P, Q)
torch.cuda.synchronize()
np.save(%(out)r + f".{rank}.npy", np.concatenate([gt.cpu().numpy().view(np.uint64).reshape(1, 48), loc.cpu().numpy().view(np.uint64).reshape(-1, 48)]))
dist.barrier(); dist.destroy_process_group()
'''


def test_sharded_world2_with_real_engine(oracle, tmp_path):
    """the N > 1 host path (bn_amd.distributed: shards, the one all_gather, world-1 products, single final exponentiation) with the
    HIP engine under it: two processes share the GPU and rendezvous over gloo (RCCL wants one GPU per rank)"""
    import os, pathlib, subprocess, sys
    from bn_amd import distributed as D
    root = pathlib.Path(__file__).resolve().parents[1]
    n = 300
    script = tmp_path / "worker.py"
    script.write_text(GPU_WORKER % {"root": str(root), "n": n, "out": str(tmp_path / "res")})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    k1 = D.synthetic_scalars(0, n, 0); k2 = D.synthetic_scalars(0, n, 1)
    P = oracle.g1_mul_batch_jacobian(np.tile(oracle.g1_one(), (n, 1)), k1); Q = oracle.g2_mul_batch_jacobian(np.tile(oracle.g2_one(), (n, 1)), k2)
    got = [np.load(str(tmp_path / f"res.{r}.npy")) for r in range(2)]
    want = oracle.pairing_product(P, Q)
    assert np.array_equal(got[0][0], want) and np.array_equal(got[1][0], want)
    assert np.array_equal(np.concatenate([got[0][1:], got[1][1:]]), oracle.pairing_batch(P, Q))


def test_bench_multi_rank_control_flow(tmp_path):
    """bench.py --gpus 2 started as a PLAIN python command (no torch.distributed.run around it): it re-launches itself, shards
    BASELINE configs[2] (2^20 total -> 2^19 per rank; shrunk here with --batch) and rank 0 prints one JSON line"""
    import json, os, pathlib, subprocess, sys
    root = pathlib.Path(__file__).resolve().parents[1]
    env = dict(os.environ, BN254_BENCH_SHARE_GPU="1", BN254_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8192"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["peak"] > 0 and "frac" in d["roofline"]


@pytest.mark.parametrize("workload", ["pairing", "product"])
def test_bench_multi_c_mode(workload):
    """`bench.py --mode multi_c`: ONE host process drives N ranks through bn254_*_multi of the C ABI (here two ranks on the one GPU);
    the line says what it measured (PCIe inclusive), carries the prediction for the real multi-GPU run and the ranks' NUMA nodes"""
    import json, os, pathlib, subprocess, sys
    root = pathlib.Path(__file__).resolve().parents[1]
    env = dict(os.environ, BN254_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(root / "bench.py"), "--mode", "multi_c", "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload]
    if workload == "pairing":
        cmd += ["--batch", "4096"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["mode"] == "multi_c" and d["pcie_inclusive"] is True and d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["devices"] == [0, 0] and d["config"]["exchange"] == "peer" and len(d["config"]["rank_numa_nodes"]) == 2
    if workload == "product":
        es = d["expected_scaling_kernels_only"]
        assert es["speedup_vs_1_gpu"] > 1.5
        # the prediction is DERIVED from a committed one-GPU line (no literals in bench.py): it names the file and reproduces from it
        src = json.loads((root / es["inputs_from"]).read_text().splitlines()[0])["side"]
        assert abs(es["ms_per_step"] - (src["product_2_17"]["ms_per_step"] + 0.05)) < 1e-9
        assert abs(es["speedup_vs_1_gpu"] - src["product_2_18"]["ms_per_step"] / es["ms_per_step"]) < 1e-9


def test_cpp_host_drives_multi_device_entry_points(oracle, tmp_path):
    """a compiled host program (g++, no Python in the loop) on include/bn254.hpp: bn::pairing, Gt::inverse, bn::MultiGpu with two
    ranks on device 0 - pairing_batch and pairing_product equal the oracle's fold of shootout/main.rs:11-16 - and bn::PreparedG2 (batch and product)"""
    import pathlib, subprocess
    root = pathlib.Path(__file__).resolve().parents[1]
    src = tmp_path / "host.cpp"
    src.write_text(r'''
#include "bn254.hpp"
#include <cstdio>
template <class T> void dump(const T &t) { const uint64_t *w = reinterpret_cast<const uint64_t *>(&t); for (size_t i = 0; i < sizeof(T) / 8; ++i) std::printf("%llu ", (unsigned long long)w[i]); std::printf("\n"); }
int main() {
    using namespace bn;
    std::vector<G1> p; std::vector<G2> q;
    G1 a = G1::one(); G2 b = G2::one();
    for (int i = 0; i < 5; ++i) { p.push_back(a); q.push_back(b); a = a + G1::one(); b = b + b; }      // (i+1) G1, 2^i G2: Jacobian z != 1
    p[3] = G1::zero();
    MultiGpu m({0, 0});
    std::vector<Gt> out = m.pairing_batch(p, q);
    for (auto &g : out) dump(g);
    dump(m.pairing_product(p, q));
    Gt e = pairing(p[1], q[1]);
    dump(e.inverse() * e);
    std::printf("%d\n", m.uses_rccl() ? 1 : 0);
    PreparedG2 vk(q[1]);                                   // the native prepared mode from C++: precompute once, pair many
    for (auto &g : vk.pairing_batch(p)) dump(g);
    PreparedG2 all(q);
    for (auto &g : all.pairing_batch(p)) dump(g);
    dump(all.pairing_product(p));                          // the multi-pairing over the prepared points
    return 0;
}
''')
    exe = tmp_path / "host"
    subprocess.check_call(["g++", "-std=c++17", "-I", str(root / "include"), str(src), "-o", str(exe),
                           "-L", str(root / "bn_amd"), "-lbn254_hip", "-Wl,-rpath," + str(root / "bn_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    lines = subprocess.check_output([str(exe)], timeout=600).decode().strip().split("\n")
    got = [np.array([int(x) for x in l.split()], np.uint64) for l in lines[:7]]
    P = np.stack([oracle.g1_one()] * 5); Q = np.stack([oracle.g2_one()] * 5)
    for i in range(1, 5):
        P[i] = oracle.g1_add(P[i - 1], oracle.g1_one()); Q[i] = oracle.g2_add(Q[i - 1], Q[i - 1])
    P[3] = oracle.g1_zero()
    want = oracle.pairing_batch(P, Q)
    for i in range(5):
        assert np.array_equal(got[i], want[i]), i
    assert np.array_equal(got[5], oracle.pairing_product(P, Q))
    assert np.array_equal(got[6], oracle.fq12_one())
    assert lines[7].strip() == "0"                     # two ranks on one device: peer-copy exchange, not RCCL
    more = [np.array([int(x) for x in l.split()], np.uint64) for l in lines[8:18]]
    assert np.array_equal(np.stack(more[:5]), oracle.pairing_batch(P, np.tile(Q[1], (5, 1))))          # bn::PreparedG2 of one point
    assert np.array_equal(np.stack(more[5:]), want)                                                    # ... and of one point per pairing
    last = np.array([int(x) for x in lines[18].split()], np.uint64)
    assert np.array_equal(last, oracle.pairing_product(P, Q))                                          # PreparedG2::pairing_product


def test_config3_whole_2_20_on_one_gpu(oracle):
    """BASELINE.json configs[2] in full on ONE GPU: 2^20 independent pairings in one call - a 4096-index sample against the oracle
    and the eight 2^17 shards an 8-GPU run would compute (separate launches) concatenate to the same bytes"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    te = D.TorchEngine(bn_amd.Engine(0), dev)
    n = 1 << 20
    P, Q = D.synthetic_points(te, 0, n)
    out = te.pairing_batch(P, Q)
    for g in (0, 5, 7):                                        # three of the eight shards, as rank g would run them
        lo, hi = D.shard_range(n, g, 8)
        assert torch.equal(te.pairing_batch(P[lo:hi].contiguous(), Q[lo:hi].contiguous()), out[lo:hi])
    torch.cuda.synchronize()
    idx = np.sort(np.random.default_rng(13).choice(n, 4096, replace=False))
    Pn = P.cpu().numpy().view(np.uint64)[idx]; Qn = Q.cpu().numpy().view(np.uint64)[idx]
    assert np.array_equal(out.cpu().numpy().view(np.uint64)[idx], oracle.pairing_batch(Pn, Qn))


def test_full_size_bilinearity_on_gpu():
    """groups/mod.rs:798-823 (test_binlinearity) at BASELINE size, entirely on the GPU and with no oracle in the loop: for 2^16 random
    (P, Q, s): e(sP, Q) == e(P, sQ) == e(P, Q)^s - three independent kernels (GLV G1 multiplication, windowed G2 multiplication,
    windowed Gt::pow) must agree with the pairing kernels on every one of the 2^16 results, byte for byte"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    te = D.TorchEngine(bn_amd.Engine(0), dev)
    n = 1 << 16
    P, Q = D.synthetic_points(te, 1 << 21, (1 << 21) + n)
    s = D.synthetic_scalars_device(te, 1 << 25, (1 << 25) + n, 0)
    sP = te.g1_mul(P, s); sQ = te.g2_mul(Q, s)
    a = te.pairing_batch(sP, Q); b = te.pairing_batch(P, sQ); c = te.gt_pow(te.pairing_batch(P, Q), s)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c)
    one = torch.zeros(48, dtype=torch.int64, device=dev)
    assert not bool((a == a[0]).all())                          # not a constant output


def test_wire_stream_format_on_gpu(oracle, eng):
    """the crate's real byte stream (groups/mod.rs:143-205): infinity is the LONE byte 0, finite points are 4 + coordinates - records of
    variable length.  encode == concatenation of the oracle's records with infinities shortened to one byte; decode inverts it,
    reports a bad tag per record and leaves a truncated trailing record unconsumed"""
    rng = np.random.default_rng(301)
    n = 40
    P, Q = _points(oracle, rng, n)
    for i in (0, 7, 8, 39):
        P[i] = oracle.g1_zero()
    for i in (3, 39):
        Q[i] = oracle.g2_zero()
    for pts, enc1, encs, decs, zero, norm in ((P, oracle.g1_encode, eng.g1_encode_stream, eng.g1_decode_stream, oracle.g1_zero(), oracle.g1_normalize),
                                              (Q, oracle.g2_encode, eng.g2_encode_stream, eng.g2_decode_stream, oracle.g2_zero(), oracle.g2_normalize)):
        want = np.concatenate([(r[:1] if r[0] == 0 else r) for r in (enc1(p) for p in pts)])
        got = encs(pts)
        assert np.array_equal(got, want)
        out, st, used = decs(got)
        assert used == got.size and out.shape[0] == n and not st.any()
        for i in range(n):
            assert np.array_equal(out[i], zero if not pts[i][2 * len(zero) // 3:].any() else norm(pts[i]))
        # truncated tail: the last complete record boundary is reported; a bad tag costs one byte and status 3
        cut = got[:-5] if got[-1] != 0 or got.size < 2 else got
        out2, st2, used2 = decs(np.concatenate([np.array([9], np.uint8), got[:200]]))
        assert st2[0] == 3 and used2 <= 201 and out2.shape[0] >= 2 and not st2[1:].any()
        assert np.array_equal(out2[1:], out[:out2.shape[0] - 1])
        out3, st3, used3 = decs(got, max_points=5)
        assert out3.shape[0] == 5 and np.array_equal(out3, out[:5])
        # the crate's own behaviour on request (BN254_OPT_STREAM_STOP_AT_ERROR): its Decodable returns Err at the first bad record
        # (groups/mod.rs:165-175) - a good record, then a bad tag: the call stops WITH the bad record, the rest stays unconsumed
        first = got[:1] if got[0] == 0 else got[:len(enc1(pts[1]))]
        bad = np.concatenate([first, np.array([9], np.uint8), got[first.size:first.size + 300]])
        with eng.options(stream_stop_at_error=1):
            out4, st4, used4 = decs(bad)
        assert out4.shape[0] == 2 and st4[0] == 0 and st4[1] == 3 and used4 == first.size + 1
        out5, st5, used5 = decs(bad)                                   # default: every record, every status
        assert out5.shape[0] > 2 and st5[1] == 3 and not st5[2:].any() and used5 > used4


@pytest.mark.parametrize("workload", ["pairing", "product"])
def test_bench_through_rccl_process_group_at_world_1(workload):
    """First contact with RCCL through torch.distributed, on the one GPU there is: bench.py with BN254_BENCH_FORCE_DIST=1 runs
    init_process_group("nccl", device_id), barrier(device_ids), all_reduce(MAX) on a DEVICE tensor and - for the product - the
    all_gather_into_tensor of the 384-byte partial through RCCL (bn_amd.distributed no longer short-circuits a one-rank group),
    i.e. every torch.distributed call of the 8-GPU SCALE run except a transfer between two different GPUs"""
    import json, os, pathlib, subprocess, sys
    root = pathlib.Path(__file__).resolve().parents[1]
    env = dict(os.environ, BN254_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BN254_BENCH_BACKEND", "BN254_BENCH_SHARE_GPU"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "2", "--warmup", "1", "--workload", workload,
                          "--no-side", "--no-host-api", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["process_group"] == "nccl" and d["n_gpus"] == 1 and d["value"] > 0


def test_rccl_all_gather_of_partials_at_world_1(oracle):
    """bn_amd.distributed.pairing_product_sharded with a REAL one-rank RCCL group: the partial travels through
    all_gather_into_tensor on device memory, then the one-launch tail; equal to the oracle's fold"""
    import os, socket
    import torch
    import torch.distributed as dist
    import bn_amd
    from bn_amd import distributed as D
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    try:
        te = D.TorchEngine(bn_amd.Engine(0), dev)
        P, Q = D.synthetic_points(te, 40, 40 + 77)
        part = te.miller_product(P, Q)
        parts = D.all_gather_partials(part)
        assert parts.shape == (1, 48) and parts.is_cuda and torch.equal(parts[0], part)
        got = D.pairing_product_sharded(te, P, Q)
        dist.barrier(device_ids=[0])
        torch.cuda.synchronize()
        Pn = P.cpu().numpy().view(np.uint64); Qn = Q.cpu().numpy().view(np.uint64)
        assert np.array_equal(got.cpu().numpy().view(np.uint64), oracle.pairing_product(Pn, Qn))
    finally:
        dist.destroy_process_group()
        for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE"):
            os.environ.pop(k, None)


def test_multi_device_c_abi_at_config_sizes(oracle):
    """the C entry points a Rust/C++ host calls, at BASELINE sizes with EIGHT ranks (device list [0]*8: eight contexts, eight host
    threads, eight shards on the one GPU): bn254_pairing_product_multi on configs[3]'s 2^18 pairs equals the single-context product,
    bn254_pairing_batch_multi on 2^17 pairs equals the single-context batch and a 2048-index sample equals the oracle"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    te = D.TorchEngine(bn_amd.Engine(0), dev)
    n = 1 << 18
    P, Q = D.synthetic_points(te, 0, n)
    want_prod = D.pairing_product_sharded(te, P, Q)
    nb = 1 << 17
    want_batch = te.pairing_batch(P[:nb].contiguous(), Q[:nb].contiguous())
    torch.cuda.synchronize()
    Pn = P.cpu().numpy().view(np.uint64); Qn = Q.cpu().numpy().view(np.uint64)
    m = bn_amd.MultiEngine([0] * 8)
    assert m.exchange == "peer"
    assert np.array_equal(m.pairing_product(Pn, Qn), want_prod.cpu().numpy().view(np.uint64))
    got = m.pairing_batch(Pn[:nb], Qn[:nb])
    assert np.array_equal(got, want_batch.cpu().numpy().view(np.uint64))
    idx = np.sort(np.random.default_rng(17).choice(nb, 2048, replace=False))
    assert np.array_equal(got[idx], oracle.pairing_batch(Pn[idx], Qn[idx]))
    m.close()


def test_scratch_is_bounded_by_one_round(oracle):
    """bn254_pairing_batch_dev on 2^22 pairings: the context's tables are sized for ONE machine round (256 pairings per CU), not for
    the batch - round 2 allocated 4 KB per pairing (17 GB here).  Device memory taken by the call stays below 1 GB and the result
    equals the shard-wise one; Gt::pow likewise reuses one bounded window table"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    te = D.TorchEngine(bn_amd.Engine(0), dev)
    n = 1 << 22
    base = 1 << 18
    P0, Q0 = D.synthetic_points(te, 0, base)
    P = P0.repeat(n // base, 1); Q = Q0.repeat(n // base, 1)            # 2^22 inputs from 2^18 distinct pairs (the point is the size)
    out = te.empty(n, 48)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    te.pairing_batch(P, Q, out)
    torch.cuda.synchronize()
    used = free0 - torch.cuda.mem_get_info(dev)[0]
    assert used < (1 << 30), f"context scratch grew by {used / 2**20:.0f} MiB"
    ref = te.pairing_batch(P0, Q0)
    for rep in (0, 7, 15):
        assert torch.equal(out[rep * base:(rep + 1) * base], ref)
    torch.cuda.synchronize()
    idx = np.sort(np.random.default_rng(19).choice(base, 512, replace=False))
    Pn = P0.cpu().numpy().view(np.uint64)[idx]; Qn = Q0.cpu().numpy().view(np.uint64)[idx]
    assert np.array_equal(ref.cpu().numpy().view(np.uint64)[idx], oracle.pairing_batch(Pn, Qn))
    k = D.synthetic_scalars_device(te, 1 << 24, (1 << 24) + base, 0)
    free1 = torch.cuda.mem_get_info(dev)[0]
    pw = te.gt_pow(ref, k)
    torch.cuda.synchronize()
    assert free1 - torch.cuda.mem_get_info(dev)[0] < (1 << 30)
    small = te.gt_pow(ref[:300].contiguous(), k[:300].contiguous())
    assert torch.equal(pw[:300], small)


def test_gt_pow_mixed_subgroup_membership(oracle, eng):
    """Gt::pow chooses its chain per WAVE on the device (cyclotomic signed windows when every element of the wave is a pairing
    value, the general chain otherwise): a batch of 200 pairing values with two arbitrary Fq12 elements in the middle of different
    waves equals the oracle everywhere"""
    rng = np.random.default_rng(206)
    n = 200
    P, Q = _points(oracle, rng, n)
    g = eng.pairing_batch(P, Q)
    g[17] = oracle.miller_only(P[17], Q[17]); g[150] = oracle.miller_only(P[150], Q[150])          # off the cyclotomic subgroup
    s = _fr(oracle, _scalars(rng, n))
    pw = eng.gt_pow_batch(g, s)
    for i in list(range(0, n, 7)) + [16, 17, 18, 149, 150, 151]:
        assert np.array_equal(pw[i], oracle.gt_pow(g[i], s[i])), i


@pytest.mark.parametrize("m", [2, 4])
def test_shared_accumulator_miller_kernels(oracle, m):
    """bn254_miller_shared{2,4}_B (m pairs per lane pair on one accumulator; chosen by the host from two / four machine rounds of pairs
    on, forced here by BN254_OPT_MILLER_SHARED): ragged sizes (not a multiple of m), infinite pairs, and 5000 pairs against the oracle's
    fold; equal to the plain path bit for bit"""
    import os
    import bn_amd
    rng = np.random.default_rng(207 + m)
    n = 5003
    te_p, te_q = _points(oracle, rng, 64)
    reps = (n + 63) // 64
    P = np.tile(te_p, (reps, 1))[:n].copy(); Q = np.tile(te_q, (reps, 1))[:n].copy()       # 64 distinct pairs, repeated
    P[7] = oracle.g1_zero(); Q[4000] = oracle.g2_zero(); P[n - 1] = oracle.g1_zero()
    e = bn_amd.Engine(0)
    # 5003 pairs would otherwise run one per wave (BN254_OPT_WAVE_PAIRING_MAX) or four lanes per pairing (BN254_OPT_QUAD_MAX)
    with e.options(miller_shared=1, wave_pairing_max=0, quad_max=0):
        plain = e.pairing_product(P, Q)
        e.set_option("miller_shared", m)
        e.profile(True); e.profile_reset()
        shared = e.pairing_product(P, Q)
        assert e.kernel_stats("miller_shared")[1] >= 1 and e.kernel_stats("miller")[1] == 0
        e.profile(False)
    assert e.get_option("miller_shared") == 0
    assert np.array_equal(plain, shared)
    assert np.array_equal(shared, oracle.pairing_product(P, Q))
    e.close()


def _gpu_count():
    import torch
    return torch.cuda.device_count()


def test_rccl_exchange_between_distinct_gpus(oracle):
    """the one branch no single-GPU box can run: bn254_multi_* over DISTINCT devices (ncclCommInitAll over the device list, grouped
    ncclAllGather of the 384-byte partials, one final exponentiation).  Skipped unless the box has two GPUs."""
    if _gpu_count() < 2:
        pytest.skip("needs two GPUs")
    import os
    import bn_amd
    rng = np.random.default_rng(401)
    n = 301
    P, Q = _points(oracle, rng, n)
    P[7] = oracle.g1_zero(); Q[9] = oracle.g2_zero()
    want_b = oracle.pairing_batch(P, Q); want_p = oracle.pairing_product(P, Q)
    devs = list(range(min(_gpu_count(), 8)))
    for kind in ("rccl", "peer"):
        m = bn_amd.MultiEngine(devs, exchange=kind)                  # bn254_multi_create_ex: forced, no fall-back
        assert m.exchange == kind
        assert np.array_equal(m.pairing_batch(P, Q), want_b), kind
        for _ in range(3):
            assert np.array_equal(m.pairing_product(P, Q), want_p), kind
        assert np.array_equal(m.pairing_product(P[:1], Q[:1]), oracle.pairing_product(P[:1], Q[:1]))
        m.close()


NCCL_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [%(root)r, %(root)r + "/oracle", %(root)r + "/tests"]
import bn_amd
from bn_amd import distributed as D
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", rank)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
eng = D.TorchEngine(bn_amd.Engine(rank), dev)
n = %(n)d
lo, hi = D.shard_range(n, rank, world)
P, Q = D.synthetic_points(eng, lo, hi)
gt = D.pairing_product_sharded(eng, P, Q)
torch.cuda.synchronize()
np.save(%(out)r + f".{rank}.npy", gt.cpu().numpy().view(np.uint64).reshape(1, 48))
dist.barrier(); dist.destroy_process_group()
'''


def test_sharded_product_over_rccl_world2(oracle, tmp_path):
    """one process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI): shards, all_gather_into_tensor of the partials,
    single final exponentiation - bit-exact on both ranks.  Skipped unless the box has two GPUs."""
    if _gpu_count() < 2:
        pytest.skip("needs two GPUs")
    import os, pathlib, subprocess, sys
    from bn_amd import distributed as D
    root = pathlib.Path(__file__).resolve().parents[1]
    n = 300
    script = tmp_path / "worker.py"
    script.write_text(NCCL_WORKER % {"root": str(root), "n": n, "out": str(tmp_path / "res")})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29549", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    k1 = D.synthetic_scalars(0, n, 0); k2 = D.synthetic_scalars(0, n, 1)
    P = oracle.g1_mul_batch_jacobian(np.tile(oracle.g1_one(), (n, 1)), k1); Q = oracle.g2_mul_batch_jacobian(np.tile(oracle.g2_one(), (n, 1)), k2)
    want = oracle.pairing_product(P, Q)
    for r in range(2):
        assert np.array_equal(np.load(str(tmp_path / f"res.{r}.npy"))[0], want), r


def test_gt_pow_modes(oracle, eng):
    """Gt::pow's three chains (BN254_OPT_GT_POW_MODE): the Frobenius decomposition (default; needs order r, which every value of the
    reference's Gt type has), the one-dimensional cyclotomic chain and the general chain agree with fields/mod.rs:35-46 on pairing
    values; a cyclotomic element of another order (the easy part of the final exponentiation applied to an arbitrary Fq12 - not
    constructible through the reference's API) is exact in the strict and general modes"""
    import os
    rng = np.random.default_rng(402)
    n = 70
    P, Q = _points(oracle, rng, n)
    g = eng.pairing_batch(P, Q)
    g[3] = oracle.fq12_one()
    sv = _scalars(rng, n); sv[:6] = [0, 1, M.R_ORD - 1, M.R_ORD - 2, (1 << 253) + 11, 6 * M.U * M.U % M.R_ORD]
    s = _fr(oracle, sv)
    want = np.stack([oracle.gt_pow(g[i], s[i]) for i in range(n)])
    try:
        for mode in (0, 2, 1):
            eng.set_option("gt_pow_mode", mode)
            assert eng.get_option("gt_pow_mode") == mode
            assert np.array_equal(eng.gt_pow_batch(g, s), want), mode
        # 20 000 pairing values ^ distinct random scalars: the Frobenius chain against the one-dimensional one, all of them
        import torch
        from bn_amd import distributed as D
        te = D.TorchEngine(eng, torch.device("cuda", 0))
        big = 20000
        Pd, Qd = D.synthetic_points(te, 777, 777 + big)
        gd = te.pairing_batch(Pd, Qd)
        kd = D.synthetic_scalars_device(te, 1 << 25, (1 << 25) + big, 0)
        eng.set_option("gt_pow_mode", 0); a = te.gt_pow(gd, kd); torch.cuda.synchronize()
        eng.set_option("gt_pow_mode", 2); b = te.gt_pow(gd, kd); torch.cuda.synchronize()
        assert torch.equal(a, b)
        raw = oracle.miller_only(P[0], Q[0])
        # f^(q^6 - 1) then ^(q^2 + 1): cyclotomic, but of order dividing (q^4 - q^2 + 1), not r
        cyc = oracle.fq12_final_exp_first_chunk(raw)              # fq12.rs:41-52
        batch = np.tile(cyc, (4, 1))
        for mode in (2, 1):
            eng.set_option("gt_pow_mode", mode)
            got = eng.gt_pow_batch(batch, s[6:10])
            for i in range(4):
                assert np.array_equal(got[i], oracle.gt_pow(cyc, s[6 + i])), (mode, i)
        # the option is validated: 3 is no mode
        with pytest.raises(Exception):
            eng.set_option("gt_pow_mode", 3)
    finally:
        eng.set_option("gt_pow_mode", None)
        assert eng.get_option("gt_pow_mode") == 0

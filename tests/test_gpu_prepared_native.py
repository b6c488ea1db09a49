"""GPU parity tests of the NATIVE prepared-G2 mode (include/bn254_hip.h bn254_g2_prepare / bn254_pairing_prepared_native_batch):
the device-native counterpart of the reference's G2Precomp (groups/mod.rs:472-520,557-588) against the CPU oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import bn_model as M
from bn_oracle import FQ, FR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    """an engine whose SMALL prepared calls run the native kernels too (by default calls of up to 12 x CUs pairings are served by the general path's
    one-pairing-per-wave kernels on the points kept with the handle: test_small_calls_are_routed_to_the_wave_kernels)"""
    import bn_amd
    e = bn_amd.Engine(0)
    e.set_option("wave_pairing_max", 0)
    return e


def _scalars(rng, n):
    return [int.from_bytes(rng.bytes(64), "little") % M.R_ORD for _ in range(n)]


def _fr(oracle, vals):
    return np.stack([oracle.fp_from_int(FR, v) for v in vals])


def _g1(oracle, vals):
    return oracle.g1_mul_batch_jacobian(np.tile(oracle.g1_one(), (len(vals), 1)), _fr(oracle, vals))


def _g2(oracle, vals):
    return oracle.g2_mul_batch_jacobian(np.tile(oracle.g2_one(), (len(vals), 1)), _fr(oracle, vals))


def test_reference_known_answer_through_the_native_table(oracle, kats, eng):
    """groups/mod.rs:773-796 (test_reduced_pairing): pairing(k1 G1, k2 G2) with Q = k2 G2 prepared natively equals the reference's twelve
    field elements; and the Q of test_prepared_g2 (groups/mod.rs:637-762) paired natively equals pairing() on the same inputs"""
    k = kats["test_reduced_pairing"]
    P = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_decimal(FR, k["k1"])); Q = oracle.g2_mul(oracle.g2_one(), oracle.fp_from_decimal(FR, k["k2"]))
    prep = eng.g2_prepare(Q)
    assert prep.count == 1 and prep.device_bytes >= 33792 + 4 + 192          # table, infinity flag, the point itself (repeated for small calls)
    assert oracle.fq12_to_ints(eng.pairing_prepared_native_batch(P, prep)[0]) == [int(x) for x in k["expected"]]
    Q2 = oracle.g2_mul(oracle.g2_one(), oracle.fp_from_decimal(FR, kats["test_prepared_g2"]["k2"]))
    prep2 = eng.g2_prepare(Q2)
    assert np.array_equal(eng.pairing_prepared_native_batch(P, prep2)[0], oracle.pairing(P, Q2))
    prep.close(); prep2.close()


def test_shared_q_ragged_batch_matches_oracle(oracle, eng):
    """200 P (ragged: not a multiple of the 32 pairings of a wave) against ONE prepared Q: random Jacobian points, the generator with z = 1,
    small and edge scalars, points at infinity - every output equals the oracle's pairing()"""
    rng = np.random.default_rng(601)
    n = 200
    Q = _g2(oracle, _scalars(rng, 1))[0]
    vals = _scalars(rng, n)
    vals[:6] = [1, 2, 3, M.R_ORD - 1, M.R_ORD - 2, 1 << 200]
    P = _g1(oracle, vals)
    P[7] = oracle.g1_one()                    # affine input, z = 1
    P[11] = oracle.g1_zero(); P[n - 1] = oracle.g1_zero()
    prep = eng.g2_prepare(Q)
    got = eng.pairing_prepared_native_batch(P, prep)
    assert np.array_equal(got, oracle.pairing_batch(P, np.tile(Q, (n, 1))))
    assert np.array_equal(got[11], oracle.fq12_one()) and np.array_equal(got[n - 1], oracle.fq12_one())
    # a prepared point at infinity: every pairing is one (groups/mod.rs:766)
    pz = eng.g2_prepare(oracle.g2_zero())
    one = eng.pairing_prepared_native_batch(P[:5], pz)
    assert np.array_equal(one, oracle.pairing_batch(P[:5], np.tile(oracle.g2_zero(), (5, 1))))
    prep.close(); pz.close()


def test_one_table_per_pairing_matches_oracle(oracle, eng):
    """per-P coefficients: 200 distinct Q prepared in ONE handle, p[i] against Q[i]; infinity on either side; q_first offsets through the
    device entry point; wrong counts are rejected"""
    import torch
    import bn_amd
    from bn_amd import _native
    from bn_amd import distributed as D
    rng = np.random.default_rng(602)
    n = 200
    P = _g1(oracle, _scalars(rng, n)); Q = _g2(oracle, _scalars(rng, n))
    Q[3] = oracle.g2_zero(); P[9] = oracle.g1_zero(); Q[n - 1] = oracle.g2_one()
    prep = eng.g2_prepare(Q)
    assert prep.count == n
    want = oracle.pairing_batch(P, Q)
    assert np.array_equal(eng.pairing_prepared_native_batch(P, prep), want)
    # fewer P than points: the first ones
    assert np.array_equal(eng.pairing_prepared_native_batch(P[:70], prep), want[:70])
    # more P than points: rejected, nothing runs
    with pytest.raises(_native.Bn254Error):
        eng.pairing_prepared_native_batch(np.concatenate([P, P[:1]]), prep)
    # q_first: p[i] against point 50 + i
    dev = torch.device("cuda", 0)
    te = D.TorchEngine(eng, dev)
    dp = torch.from_numpy(P[50:150].view(np.int64)).to(dev)
    out = te.empty(100, 48)
    eng.pairing_prepared_native_dev(dp.data_ptr(), prep, out.data_ptr(), 100, q_first=50, stream=te._stream())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want[50:150])
    with pytest.raises(_native.Bn254Error):
        eng.pairing_prepared_native_dev(dp.data_ptr(), prep, out.data_ptr(), 100, q_first=150, stream=te._stream())
    prep.close()


def test_table_is_a_function_of_the_point_alone(oracle, eng, kats):
    """the native table holds canonical field elements: two Jacobian representations of one point (z = 1 and z != 1) give byte-identical
    tables; the device's table equals the host simulation's (the same templates compiled for the CPU with every bound enforced); and every
    record holds canonical limbs, B and xi B in the (u, v) operand form with xi B = (9 + i) B"""
    import hostsim_lib
    Qj = oracle.g2_mul(oracle.g2_one(), oracle.fp_from_decimal(FR, kats["test_prepared_g2"]["k2"]))
    Qa = oracle.g2_normalize(Qj)
    assert not np.array_equal(Qa, Qj)
    pj = eng.g2_prepare(Qj); pa = eng.g2_prepare(Qa)
    tj = pj.export(); ta = pa.export()
    assert tj.shape == (88, 12, 2, 4) and np.array_equal(tj, ta)
    hs = hostsim_lib.HostSim(bounds=True)
    U32 = C.POINTER(C.c_uint32)
    sim = np.zeros((88, 2, 48), np.uint32)
    hs.lib.hsb_native_precompute(np.ascontiguousarray(Qj).ctypes.data_as(U32), sim.ctypes.data_as(U32))
    dev_tab = tj.transpose(0, 2, 1, 3).reshape(88, 2, 48)            # [line][lane][12 groups x 4 words]
    assert np.array_equal(dev_tab, sim)
    # limbs -> integers (radix 2^29, Montgomery radix 2^261)
    rinv = pow(1 << 261, -1, M.Q)
    def val(w):
        return sum(int(x) << (29 * i) for i, x in enumerate(w)) * rinv % M.Q
    for line in (0, 1, 40, 87):
        even, odd = dev_tab[line, 0], dev_tab[line, 1]
        for w in (even, odd):
            for off in (0, 9, 18, 28, 37):
                assert sum(int(x) << (29 * i) for i, x in enumerate(w[off:off + 9])) < M.Q           # canonical
        # a prepared multiplier (u, v): this lane's component of z * b is own_z * u + partner_z * v, i.e. u = b0 in both lanes, v = -b1 | b1
        def unprep(off):
            b0, b1 = val(even[off:off + 9]), val(odd[off + 9:off + 18])
            assert val(odd[off:off + 9]) == b0 and val(even[off + 9:off + 18]) == (-b1) % M.Q
            return b0, b1
        b = unprep(28); xb = unprep(9)
        assert xb == ((9 * b[0] - b[1]) % M.Q, (9 * b[1] + b[0]) % M.Q)                              # xi B = (9 + i) B
    pj.close(); pa.close()


def test_full_size_shared_and_per_pairing(oracle):
    """BASELINE configs[1] size: 2^16 P against one prepared Q and 2^16 (P, Q) pairs with one table each - the whole batches equal the fused
    kernels' pairing_batch (an independent device path: lines computed on the fly, other schedule bookkeeping, other line normalisation) and an
    oracle sample; sub-launches (round of 2^14) return the same bytes"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    n = 1 << 16
    e = bn_amd.Engine(0)
    te = D.TorchEngine(e, dev)
    P, Q = D.synthetic_points(te, 0, n)
    # one table per pairing (2.2 GB of tables)
    prep = e.g2_prepare_dev(Q.data_ptr(), n, te._stream())
    assert prep.count == n and prep.device_bytes == (n + 1) * 33792 + n * (4 + 192)       # the tables, the identity record, flags, the points
    out = te.empty(n, 48)
    e.pairing_prepared_native_dev(P.data_ptr(), prep, out.data_ptr(), n, stream=te._stream())
    ref = D.pairing_batch_sharded(te, P, Q)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    idx = np.random.default_rng(11).choice(n, 512, replace=False)
    Pn = P.cpu().numpy().view(np.uint64); Qn = Q.cpu().numpy().view(np.uint64)
    assert np.array_equal(out.cpu().numpy().view(np.uint64)[idx], oracle.pairing_batch(Pn[idx], Qn[idx]))
    with e.options(round_pairs=1 << 14):
        out2 = te.empty(n, 48)
        e.pairing_prepared_native_dev(P.data_ptr(), prep, out2.data_ptr(), n, stream=te._stream())
        torch.cuda.synchronize()
        assert torch.equal(out2, out)
    prep.close()
    # one shared Q
    q0 = Q[12345:12346].contiguous()
    prep1 = e.g2_prepare_dev(q0.data_ptr(), 1, te._stream())
    e.pairing_prepared_native_dev(P.data_ptr(), prep1, out.data_ptr(), n, stream=te._stream())
    ref1 = D.pairing_batch_sharded(te, P, q0.expand(n, 24).contiguous())
    torch.cuda.synchronize()
    assert torch.equal(out, ref1)
    assert np.array_equal(out.cpu().numpy().view(np.uint64)[idx], oracle.pairing_batch(Pn[idx], np.tile(Qn[12345], (len(idx), 1))))
    prep1.close()
    e.close()


def test_multi_device_entry_points_of_the_native_mode(oracle):
    """bn254_g2_prepare_multi / bn254_pairing_prepared_native_batch_multi with two and three ranks on the one GPU: one shared point prepared on
    every rank (ragged n), and a sharded set of points (p[i] against point i) - equal to the oracle; mismatched counts are rejected"""
    import bn_amd
    from bn_amd import _native
    rng = np.random.default_rng(603)
    n = 101
    P = _g1(oracle, _scalars(rng, n)); Q = _g2(oracle, _scalars(rng, n))
    P[5] = oracle.g1_zero(); Q[50] = oracle.g2_zero()
    want = oracle.pairing_batch(P, Q)
    for devs in ([0, 0], [0, 0, 0]):
        m = bn_amd.MultiEngine(devs)
        m.set_option("wave_pairing_max", 0)                                # the native kernels, not the small-call route
        one = m.g2_prepare(Q[1])
        assert one.count == 1
        assert np.array_equal(m.pairing_prepared_native_batch(P, one), oracle.pairing_batch(P, np.tile(Q[1], (n, 1))))
        allq = m.g2_prepare(Q)
        assert allq.count == n
        assert np.array_equal(m.pairing_prepared_native_batch(P, allq), want)
        with pytest.raises(_native.Bn254Error):
            m.pairing_prepared_native_batch(P[:50], allq)                  # a sharded set pairs with exactly as many points
        # the multi-pairing over the prepared points: shards fold locally (shared-accumulator kernels at this size via round_pairs), one exchange
        m.set_option("round_pairs", 8)
        assert np.array_equal(m.pairing_product_prepared_native(P, allq), oracle.pairing_product(P, Q))
        assert np.array_equal(m.pairing_product_prepared_native(P[:77], one), oracle.pairing_product(P[:77], np.tile(Q[1], (77, 1))))
        assert np.array_equal(m.pairing_product_prepared_native(P[:0], one), oracle.fq12_one())
        with pytest.raises(_native.Bn254Error):
            m.pairing_product_prepared_native(P[:50], allq)
        one.close(); allq.close(); m.close()


def test_python_mirror_of_the_prepared_mode(oracle):
    """bn_amd.PreparedG2 (bn_amd/api.py): the crate-style objects - pairing(p, q) == PreparedG2(q).pairing(p), bilinear in p"""
    import bn_amd
    rng = np.random.default_rng(604)
    p = bn_amd.G1.random(rng); q = bn_amd.G2.random(rng); s = bn_amd.Fr.random(rng)
    vk = bn_amd.PreparedG2(q)
    assert len(vk) == 1
    assert vk.pairing(p) == bn_amd.pairing(p, q)
    assert vk.pairing(p * s) == bn_amd.pairing(p, q).pow(s)
    assert vk.pairing(bn_amd.G1.zero()) == bn_amd.Gt.one()
    both = bn_amd.PreparedG2([q, q * s])
    got = both.pairing_batch([p, p])
    assert bn_amd.Gt(got[0]) == bn_amd.pairing(p, q) and bn_amd.Gt(got[1]) == bn_amd.pairing(p, q).pow(s)
    vk.close(); both.close()


def test_committed_goldens_through_the_native_tables(goldens, eng):
    """tests/golden/pairing_goldens.npz (96 pairings with edge scalars, made by the KAT-pinned oracle): every (g1, g2) pair through a native
    table of its own g2, and the first g2 as ONE shared table against the pairs that use it"""
    g1, g2, gt = goldens["g1"], goldens["g2"], goldens["gt"]
    prep = eng.g2_prepare(g2)
    assert np.array_equal(eng.pairing_prepared_native_batch(g1, prep), gt)
    prep.close()
    same = [i for i in range(len(g2)) if np.array_equal(g2[i], g2[0])]
    one = eng.g2_prepare(g2[0])
    assert np.array_equal(eng.pairing_prepared_native_batch(g1[same], one), gt[same])
    one.close()


def test_small_calls_are_routed_to_the_wave_kernels(oracle):
    """default options: a prepared call of up to 12 x CUs pairings runs the general path's one-pairing-per-wave kernel on the points kept with the
    handle (1.0 ms instead of the 1.7 ms of a lane-pair Miller loop), a larger one the native kernels - same bytes either way, and equal to the
    same call with the route switched off"""
    import bn_amd
    rng = np.random.default_rng(605)
    n = 300
    P = _g1(oracle, _scalars(rng, n)); Q = _g2(oracle, _scalars(rng, n))
    P[2] = oracle.g1_zero(); Q[4] = oracle.g2_zero()
    e = bn_amd.Engine(0)
    cus = e.get_option("round_pairs") // 256
    one = e.g2_prepare(Q[0]); allq = e.g2_prepare(Q)
    e.profile(True)
    for prep, want in ((one, oracle.pairing_batch(P, np.tile(Q[0], (n, 1)))), (allq, oracle.pairing_batch(P, Q))):
        e.profile_reset()
        got = e.pairing_prepared_native_batch(P, prep)
        assert e.kernel_stats("pairing_wave")[1] == 1 and e.kernel_stats("miller_native")[1] == 0
        assert np.array_equal(got, want)
        with e.options(wave_pairing_max=0):
            e.profile_reset()
            got2 = e.pairing_prepared_native_batch(P, prep)
            assert e.kernel_stats("pairing_wave")[1] == 0 and e.kernel_stats("miller_native")[1] == 1
        assert np.array_equal(got2, want)
    # just above the route's limit: the native kernels, by default
    big = 12 * cus + 1
    Pb = np.tile(P, (big // n + 1, 1))[:big]
    e.profile_reset()
    gb = e.pairing_prepared_native_batch(Pb, one)
    assert e.kernel_stats("miller_native")[1] == 1 and e.kernel_stats("pairing_wave")[1] == 0
    assert np.array_equal(gb[:n], oracle.pairing_batch(P, np.tile(Q[0], (n, 1)))) and np.array_equal(gb[n:2 * n], gb[:n])
    one.close(); allq.close(); e.close()


def test_product_over_native_tables_matches_fold(oracle, goldens):
    """the multi-pairing over prepared points (bn254_pairing_product_prepared_native): fold(Gt::one(), acc * pairing(p[i], q[i])) of
    shootout/main.rs:11-16 with the G2 side prepared - plain native kernels (M = 1), two and four pairs per accumulator (the shared-accumulator
    kernels, reached at test sizes through round_pairs / miller_shared), ragged n (not a multiple of M), points at infinity on either side
    (the identity record), the empty product, the small-call route, q_first through the device entry point; and the committed goldens"""
    import torch
    import bn_amd
    from bn_amd import _native
    from bn_amd import distributed as D
    rng = np.random.default_rng(606)
    n = 203
    P = _g1(oracle, _scalars(rng, n)); Q = _g2(oracle, _scalars(rng, n))
    P[0] = oracle.g1_zero(); Q[5] = oracle.g2_zero(); P[6] = oracle.g1_zero(); Q[6] = oracle.g2_zero(); P[n - 1] = oracle.g1_zero(); P[9] = oracle.g1_one()
    e = bn_amd.Engine(0)
    prep = e.g2_prepare(Q)
    want = oracle.pairing_product(P, Q)
    assert np.array_equal(e.pairing_product_prepared_native(P, prep), want)                        # default: the small-call route (wave kernels)
    assert np.array_equal(e.pairing_product_prepared_native(P[:0], prep), oracle.fq12_one())        # empty product
    e.profile(True)
    for opts, kernel in ((dict(wave_pairing_max=0), "miller_native"), (dict(wave_pairing_max=0, round_pairs=64, miller_shared=2), "miller_native_shared"),
                         (dict(wave_pairing_max=0, round_pairs=32), "miller_native_shared"), (dict(wave_pairing_max=0, round_pairs=16, miller_shared=4), "miller_native_shared")):
        with e.options(**opts):
            for cnt in (n, n - 1, n - 2, n - 3, 5, 1):
                e.profile_reset()
                got = e.pairing_product_prepared_native(P[:cnt], prep)
                assert np.array_equal(got, oracle.pairing_product(P[:cnt], Q[:cnt])), (opts, cnt)
                if cnt >= 200:
                    assert e.kernel_stats(kernel)[1] >= 1 and e.kernel_stats("pairing_wave")[1] == 0, (opts, cnt)
    # more pairs than prepared points: rejected
    with pytest.raises(_native.Bn254Error):
        e.pairing_product_prepared_native(np.concatenate([P, P[:1]]), prep)
    # q_first through the device entry point, four pairs per accumulator
    dev = torch.device("cuda", 0)
    te = D.TorchEngine(e, dev)
    dp = torch.from_numpy(P[50:180].view(np.int64)).to(dev)
    part = te.empty(1, 48)
    with e.options(wave_pairing_max=0, round_pairs=16):
        e.miller_product_prepared_native_dev(dp.data_ptr(), prep, 130, part.data_ptr(), q_first=50, stream=te._stream())
        e.final_exp_batch_dev(part.data_ptr(), part.data_ptr(), 1, stream=te._stream())
        torch.cuda.synchronize()
    assert np.array_equal(part.cpu().numpy().view(np.uint64)[0], oracle.pairing_product(P[50:180], Q[50:180]))
    # ONE prepared point against many P: prod e(p_i, Q) = e(sum p_i, Q)
    one = e.g2_prepare(Q[1])
    with e.options(wave_pairing_max=0, round_pairs=32):
        got = e.pairing_product_prepared_native(P[:101], one)
    assert np.array_equal(got, oracle.pairing_product(P[:101], np.tile(Q[1], (101, 1))))
    assert np.array_equal(e.pairing_product_prepared_native(P[:101], one), got)                     # ... and through the small-call route (the point repeated)
    one.close(); prep.close()
    # committed goldens: the product of all 96 golden pairings, through tables of their own g2
    g1, g2, gt = goldens["g1"], goldens["g2"], goldens["gt"]
    want_g = gt[0]
    for v in gt[1:]:
        want_g = oracle.fq12_mul(want_g, v)
    pg = e.g2_prepare(g2)
    with e.options(wave_pairing_max=0, round_pairs=16):
        assert np.array_equal(e.pairing_product_prepared_native(g1, pg), want_g)
    pg.close(); e.close()


def test_full_size_product_over_native_tables(oracle):
    """BASELINE configs[3] shape on one GPU with the G2 side prepared: 2^18 pairs, one table each (8.9 GB), four pairs per accumulator - equal to the
    fused multi-pairing of the same inputs (an independent device path) and, on a 512-pair prefix, to the oracle's fold; 2^17 pairs (two pairs
    per accumulator) likewise"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    dev = torch.device("cuda", 0)
    n = 1 << 18
    e = bn_amd.Engine(0)
    te = D.TorchEngine(e, dev)
    P, Q = D.synthetic_points(te, 0, n)
    prep = e.g2_prepare_dev(Q.data_ptr(), n, te._stream())
    e.profile(True)
    part = te.empty(1, 48)
    for cnt, m in ((n, 4), (n // 2, 2)):
        e.profile_reset()
        e.miller_product_prepared_native_dev(P.data_ptr(), prep, cnt, part.data_ptr(), stream=te._stream())
        e.final_exp_batch_dev(part.data_ptr(), part.data_ptr(), 1, stream=te._stream())
        torch.cuda.synchronize()
        assert e.kernel_stats("miller_native_shared")[1] == (cnt // m) >> 16
        ref = D.pairing_product_sharded(te, P[:cnt].contiguous(), Q[:cnt].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(part[0], ref.reshape(-1))
    with e.options(wave_pairing_max=0, round_pairs=64):
        e.miller_product_prepared_native_dev(P.data_ptr(), prep, 512, part.data_ptr(), stream=te._stream())
        e.final_exp_batch_dev(part.data_ptr(), part.data_ptr(), 1, stream=te._stream())
        torch.cuda.synchronize()
    # the one-process-per-GPU path (bn_amd/distributed.py) at world 1: same bytes
    got = D.pairing_product_prepared_sharded(te, P, prep)
    torch.cuda.synchronize()
    assert torch.equal(got.reshape(-1), D.pairing_product_sharded(te, P, Q).reshape(-1))
    Pn = P[:512].cpu().numpy().view(np.uint64); Qn = Q[:512].cpu().numpy().view(np.uint64)
    assert np.array_equal(part.cpu().numpy().view(np.uint64)[0], oracle.pairing_product(Pn, Qn))
    prep.close(); e.close()

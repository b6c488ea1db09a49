"""GPU parity tests of the WAVE-COOPERATIVE kernels (bn_amd/csrc/wave.hpp, bn254_kernels_w.hip): one Fq12 per wave for the
latency-bound tails of the path - the single final exponentiation of a multi-pairing, the one-launch product tree with its
arrival tree across workgroups, the product-then-exponentiate tail of the sharded product.  Everything goes through the C ABI
and is compared with the CPU oracle bit for bit (run with -m gpu on an MI355X)."""
import numpy as np
import pytest

import bn_model as M
from bn_oracle import FQ

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def te():
    import torch
    import bn_amd
    from bn_amd import distributed as D
    return D.TorchEngine(bn_amd.Engine(0), torch.device("cuda", 0))


def _rand_fq12(oracle, rng, n):
    return np.stack([np.concatenate([oracle.fp_from_int(FQ, int.from_bytes(rng.bytes(40), "little") % M.Q) for _ in range(12)]) for _ in range(n)])


def _dev(te, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(te.device)


def _host(t):
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy().view(np.uint64)


def _final_exp_batch(te, vals):
    out = te.empty(vals.shape[0], 48)
    te.e.final_exp_batch_dev(vals.data_ptr(), out.data_ptr(), vals.shape[0], te._stream())
    return out


def _fold(oracle, vals):
    acc = oracle.fq12_one()
    for v in vals:
        acc = oracle.fq12_mul(acc, v)
    return acc


def test_wave_final_exponentiation_matches_oracle(oracle, kats, te):
    """fq12.rs:41-88 with one Fq12 per wave: arbitrary Fq12 inputs (the easy part maps anything non-zero into the cyclotomic
    subgroup), the reference's own Miller-loop known answer (groups/mod.rs:522-547 -> :773-796), one and several per launch"""
    rng = np.random.default_rng(301)
    vals = _rand_fq12(oracle, rng, 9)
    want = np.stack([oracle.fq12_final_exponentiation(v) for v in vals])
    for n in (1, 2, 9):
        assert np.array_equal(_host(_final_exp_batch(te, _dev(te, vals[:n]))), want[:n]), n
    one = oracle.fq12_one()
    assert np.array_equal(_host(_final_exp_batch(te, _dev(te, one.reshape(1, 48))))[0], one)
    # literal known answers on both sides, no oracle in between: the Miller value of test_miller_loop -> the Gt of test_reduced_pairing
    kat_f = oracle.fq12_from_ints(kats["test_miller_loop"]["expected"]).reshape(1, 48)
    assert te.e.get_option("wave_fe_max") >= 1
    got = _host(_final_exp_batch(te, _dev(te, kat_f)))[0]
    assert oracle.fq12_to_ints(got) == [int(x) for x in kats["test_reduced_pairing"]["expected"]]
    # in place, as pairing_batch uses it
    d = _dev(te, vals)
    te.e.final_exp_batch_dev(d.data_ptr(), d.data_ptr(), 9, te._stream())
    assert np.array_equal(_host(d), want)


def test_wave_and_lane_pair_final_exponentiation_agree(te):
    """the same 600 Miller values through both kernels (the switch-over is a host-side threshold, BN254_OPT_WAVE_FE_MAX)"""
    import torch
    from bn_amd import distributed as D
    n = 600
    P, Q = D.synthetic_points(te, 5000, 5000 + n)
    f = te.empty(n, 48)
    te.e.miller_batch_dev(P.data_ptr(), Q.data_ptr(), f.data_ptr(), n, te._stream())
    with te.e.options(wave_fe_max=0):
        a = _final_exp_batch(te, f); torch.cuda.synchronize()
    with te.e.options(wave_fe_max=4096):
        b = _final_exp_batch(te, f); torch.cuda.synchronize()
    te.e.profile(True); te.e.profile_reset()
    with te.e.options(wave_fe_max=4096):
        _final_exp_batch(te, f[:1].contiguous()); torch.cuda.synchronize()
    assert te.e.kernel_stats("final_exp_wave")[1] == 1 and te.e.kernel_stats("final_exp")[1] == 0
    te.e.profile(False)
    assert torch.equal(a, b)


def test_one_launch_product_tree_matches_fold(oracle, te):
    """bn254_gt_product_dev = fold of shootout/main.rs:11-16 over Fq12 values: lane chunks, the wave-cooperative fold of a wave's 32
    partial products, and the arrival tree across workgroups - sizes around every boundary (one pair, one wave, several waves,
    ragged tails, chunks of two and four values per lane pair)"""
    import torch
    from bn_amd import distributed as D
    rng = np.random.default_rng(302)
    small = _rand_fq12(oracle, rng, 70)
    for n in (1, 2, 3, 31, 32, 33, 64, 65, 70):
        assert np.array_equal(_host(te.gt_product(_dev(te, small[:n]))), _fold(oracle, small[:n])), n
    n = 140000                                                    # chunk 3: 46667 groups, 1459 waves, an 11-level arrival tree
    P, Q = D.synthetic_points(te, 0, n)
    vals = te.pairing_batch(P, Q)
    host = _host(vals)
    for m in (1000, 4097, 65536, 70001, n):
        got = _host(te.gt_product(vals[:m].contiguous()))
        assert np.array_equal(got, _fold(oracle, host[:m])), m
    # every shape of the tree the host could pick (bn254_hip.hip product_shape), ragged against each of its three parameters
    want = {m: _fold(oracle, host[:m]) for m in (1, 2, 33, 1000, 4097)}
    for c, L, B in ((1, 2, 0), (3, 5, 1), (2, 32, 3), (7, 12, 2), (1, 32, 5), (4, 1, 0), (2, 3, 5)):
        with te.e.options(product_chunk=c, product_per_wave=L, product_bfly=B):
            for m, w in want.items():
                assert np.array_equal(_host(te.gt_product(vals[:m].contiguous())), w), (c, L, B, m)
    # the arrival tree is a race by design (first arriver leaves, second continues): the value must not depend on who wins
    ref = te.gt_product(vals)
    for _ in range(20):
        assert torch.equal(te.gt_product(vals), ref)


def test_product_final_exp_tail(oracle, te):
    """bn254_gt_product_final_exp_dev: what rank 0 runs after the all-gather of the sharded multi-pairing (one launch up to 16 values, product tree + exponentiation above)"""
    rng = np.random.default_rng(303)
    vals = _rand_fq12(oracle, rng, 66)
    for m in (1, 2, 8, 16, 17, 64, 66):
        want = oracle.fq12_final_exponentiation(_fold(oracle, vals[:m]))
        assert np.array_equal(_host(te.product_final_exp(_dev(te, vals[:m]))), want), m


def test_wave_pairing_matches_oracle_and_lane_pair_kernels(oracle, te):
    """small batches run the WHOLE pairing per wave (bn254_pairing_W: prologue, Miller loop and final exponentiation as one program):
    against the oracle with the edge cases of groups/mod.rs:764-771 (infinity either side, z = 1), the reference's known answer,
    and the same 300 pairs through the lane-pair kernels (BN254_OPT_WAVE_PAIRING_MAX = 0) - both sides of the host's threshold"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    from bn_oracle import FR
    rng = np.random.default_rng(305)
    n = 40
    ks = [oracle.fp_from_int(FR, int.from_bytes(rng.bytes(40), "little") % M.R_ORD) for _ in range(2 * n)]
    P = oracle.g1_mul_batch_jacobian(np.tile(oracle.g1_one(), (n, 1)), np.stack(ks[:n]))
    Q = oracle.g2_mul_batch_jacobian(np.tile(oracle.g2_one(), (n, 1)), np.stack(ks[n:]))
    P[1] = oracle.g1_zero(); Q[2] = oracle.g2_zero(); P[3] = oracle.g1_zero(); Q[3] = oracle.g2_zero()
    P[4] = oracle.g1_one(); Q[4] = oracle.g2_one()
    P[5] = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_int(FR, M.R_ORD - 1))
    e = bn_amd.Engine(0)
    e.profile(True); e.profile_reset()
    got = e.pairing_batch(P, Q)
    assert e.kernel_stats("pairing_wave")[1] == 1 and e.kernel_stats("miller")[1] == 0
    e.profile(False)
    assert np.array_equal(got, oracle.pairing_batch(P, Q))
    assert np.array_equal(e.pairing_product(P, Q), oracle.pairing_product(P, Q))          # Miller per wave -> product tree -> one exponentiation
    e.close()
    m = 300
    Pd, Qd = D.synthetic_points(te, 9000, 9000 + m)
    a = te.pairing_batch(Pd, Qd); torch.cuda.synchronize()
    with te.e.options(wave_pairing_max=0, wave_fe_max=0, quad_max=0):
        b = te.pairing_batch(Pd, Qd); torch.cuda.synchronize()
    assert torch.equal(a, b)
    Pn = Pd[:8].cpu().numpy().view(np.uint64); Qn = Qd[:8].cpu().numpy().view(np.uint64)
    assert np.array_equal(a[:8].cpu().numpy().view(np.uint64), oracle.pairing_batch(Pn, Qn))
    # either side of the host's default threshold (BN254_OPT_WAVE_PAIRING_MAX: 14 per CU - thirteen workgroups per CU, several waves per SIMD)
    e2 = te.e
    thr = e2.get_option("wave_pairing_max")
    assert thr == 14 * torch.cuda.get_device_properties(0).multi_processor_count
    big = thr + 1
    Pb, Qb = D.synthetic_points(te, 20000, 20000 + big)
    e2.profile(True); e2.profile_reset()
    w = te.pairing_batch(Pb[:thr].contiguous(), Qb[:thr].contiguous()); torch.cuda.synchronize()
    assert e2.kernel_stats("pairing_wave")[1] == 1 and e2.kernel_stats("miller")[1] == 0
    e2.profile_reset()
    with e2.options(quad_max=0):                                 # (the four-lane kernels have their own test)
        l = te.pairing_batch(Pb, Qb); torch.cuda.synchronize()
    assert e2.kernel_stats("pairing_wave")[1] == 0 and e2.kernel_stats("miller")[1] == 1
    e2.profile(False)
    assert torch.equal(w, l[:thr])
    with te.e.options(wave_pairing_max=1 << 20, wave_fe_max=1 << 20):
        assert torch.equal(te.pairing_batch(Pb, Qb), l)


def test_quad_kernels_match_oracle_and_lane_pair_kernels(oracle, te, goldens):
    """four lanes per pairing (bn_amd/csrc/quad.hpp, bn254_kernels_q.hip: Miller loop + final exponentiation, picked by the host between
    BN254_OPT_WAVE_PAIRING_MAX and BN254_OPT_QUAD_MAX pairings per call): against the oracle with the edge cases of groups/mod.rs:764-771,
    the committed goldens, and bit for bit against the lane-pair kernels on both sides of both thresholds; the multi-pairing too"""
    import torch
    import bn_amd
    from bn_amd import distributed as D
    from bn_oracle import FR
    e = bn_amd.Engine(0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert e.get_option("quad_max") == 64 * cus
    # (1) a ragged small batch forced onto the quad kernels (70 quads: 4.4 waves), edge cases included
    rng = np.random.default_rng(306)
    n = 70
    ks = [oracle.fp_from_int(FR, int.from_bytes(rng.bytes(40), "little") % M.R_ORD) for _ in range(2 * n)]
    P = oracle.g1_mul_batch_jacobian(np.tile(oracle.g1_one(), (n, 1)), np.stack(ks[:n]))
    Q = oracle.g2_mul_batch_jacobian(np.tile(oracle.g2_one(), (n, 1)), np.stack(ks[n:]))
    P[1] = oracle.g1_zero(); Q[2] = oracle.g2_zero(); P[3] = oracle.g1_zero(); Q[3] = oracle.g2_zero()
    P[4] = oracle.g1_one(); Q[4] = oracle.g2_one()
    P[69] = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_int(FR, M.R_ORD - 1))
    with e.options(wave_pairing_max=0, wave_fe_max=0):
        e.profile(True); e.profile_reset()
        got = e.pairing_batch(P, Q)
        assert e.kernel_stats("miller_quad")[1] == 1 and e.kernel_stats("final_exp_quad")[1] == 1 and e.kernel_stats("miller")[1] == 0
        e.profile(False)
        assert np.array_equal(got, oracle.pairing_batch(P, Q))
        assert np.array_equal(e.pairing_product(P, Q), oracle.pairing_product(P, Q))      # Miller per quad -> product tree -> one exponentiation
        g = goldens
        assert np.array_equal(e.pairing_batch(g["g1"], g["g2"]), g["gt"])                  # the committed fixtures, no oracle involved
        assert np.array_equal(e.pairing_batch(P[:1], Q[:1]), got[:1])                      # one quad: a quarter... a sixteenth of a wave
    e.close()
    # (2) both sides of both thresholds on device-resident inputs: thr_w (wave | quad) and thr_q (quad | lane pair)
    e2 = te.e
    # (the Miller loop and the exponentiation have their own wave thresholds: above BOTH the whole pairing runs on four lanes)
    thr_w, thr_q = max(e2.get_option("wave_pairing_max"), e2.get_option("wave_fe_max")), e2.get_option("quad_max")
    Pd, Qd = D.synthetic_points(te, 40000, 40000 + thr_q + 1)
    with e2.options(quad_max=0):
        ref = te.pairing_batch(Pd, Qd); torch.cuda.synchronize()                           # lane-pair kernels (n > thr_w)
    e2.profile(True)
    for m, want_quad in ((thr_w + 1, True), (thr_q, True), (thr_q + 1, False)):
        e2.profile_reset()
        out = te.pairing_batch(Pd[:m].contiguous(), Qd[:m].contiguous()); torch.cuda.synchronize()
        assert (e2.kernel_stats("miller_quad")[1], e2.kernel_stats("final_exp_quad")[1]) == ((1, 1) if want_quad else (0, 0)), m
        assert e2.kernel_stats("miller")[1] == (0 if want_quad else 1), m
        assert torch.equal(out, ref[:m]), m
        if m == thr_q:
            # the four-lane kernels at their REAL operating size against the ORACLE itself (not only against the lane-pair kernels): 2048
            # indices spread over the whole launch - first and last waves, every XCD's share, both pairs of a quad
            idx = np.unique(np.concatenate([np.arange(64), np.arange(m - 64, m), np.random.default_rng(308).integers(0, m, 1920)]))
            ti = torch.from_numpy(idx).to(Pd.device)
            want = oracle.pairing_batch(Pd[ti].cpu().numpy().view(np.uint64), Qd[ti].cpu().numpy().view(np.uint64))
            assert np.array_equal(out[ti].cpu().numpy().view(np.uint64), want), "four-lane kernels differ from the oracle at n = quad_max"
    e2.profile(False)
    Pn = Pd[:6].cpu().numpy().view(np.uint64); Qn = Qd[:6].cpu().numpy().view(np.uint64)
    assert np.array_equal(ref[:6].cpu().numpy().view(np.uint64), oracle.pairing_batch(Pn, Qn))
    # (3) the halves separately: Miller values of the quad kernel through the lane-pair exponentiation and vice versa
    m = thr_w + 77
    with e2.options(quad_max=0):
        a = te.miller_product(Pd[:m].contiguous(), Qd[:m].contiguous()); torch.cuda.synchronize()
    b = te.miller_product(Pd[:m].contiguous(), Qd[:m].contiguous()); torch.cuda.synchronize()
    # un-exponentiated products differ by nothing: both run the NAF schedule on the isomorphic curve with the same lines
    assert torch.equal(te.product_final_exp(a.reshape(1, 48)), te.product_final_exp(b.reshape(1, 48)))


def test_single_pairing_latency_path(oracle, te):
    """n = 1 through every entry point: the by-value `pairing(p, q)` of lib.rs:181-183"""
    import bn_amd
    rng = np.random.default_rng(304)
    k = [int.from_bytes(rng.bytes(40), "little") % M.R_ORD for _ in range(2)]
    from bn_oracle import FR
    P = oracle.g1_mul(oracle.g1_one(), oracle.fp_from_int(FR, k[0])); Q = oracle.g2_mul(oracle.g2_one(), oracle.fp_from_int(FR, k[1]))
    want = oracle.pairing(P, Q)
    e = bn_amd.Engine(0)
    assert np.array_equal(e.pairing_batch(P, Q)[0], want)
    assert np.array_equal(e.pairing_product(P.reshape(1, 12), Q.reshape(1, 24)), want)
    e.close()

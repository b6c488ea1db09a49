"""CPU tests of the wave-cooperative Fq12 machine (bn_amd/csrc/wave.hpp): the generated role tables and programs against the
independent big-integer model, and the device templates themselves - executed by the host simulation on 32 threads, one per lane
pair, with every bound of the lazy number system enforced - against the oracle, bit for bit."""
import importlib.util
import pathlib
import random

import numpy as np
import pytest

import bn_model as M
import hostsim_lib
from bn_oracle import FQ

ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def gen():
    spec = importlib.util.spec_from_file_location("gen_wave_tables", ROOT / "tools" / "gen_wave_tables.py")
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    return g


@pytest.fixture(scope="module")
def hs():
    return hostsim_lib.HostSim(bounds=True)


def _flat(x):            # bn_model Fq12 ((c0,c1,c2),(c0,c1,c2)) -> the generator's list of six Fq2
    return [x[0][0], x[0][1], x[0][2], x[1][0], x[1][1], x[1][2]]


def _unflat(v):
    return ((v[0], v[1], v[2]), (v[3], v[4], v[5]))


def test_tables_against_the_big_integer_model(gen):
    """the programs the kernels interpret, executed on exact field elements, equal oracle/bn_model.py (fq12.rs formulas)"""
    B, progs = gen.build()
    gen.self_check(B, progs)                                    # the generator's own model ...
    rnd = random.Random(3)
    rf12 = lambda: [(rnd.randrange(M.Q), rnd.randrange(M.Q)) for _ in range(6)]
    def run(name, a, b=None):
        regs = gen.fresh_regs()
        if b is not None:
            for r, v in zip(gen.RES, b): regs[r] = v
            gen.run(B, progs["PUT0"], regs)
        for r, v in zip(gen.RES, a): regs[r] = v
        gen.run(B, progs[name], regs)
        return [regs[r] for r in gen.RES]
    a, b = rf12(), rf12()                                       # ... and the oracle's, independently written
    assert run("MUL", a, b) == _flat(M.f12_mul(_unflat(a), _unflat(b)))
    assert run("MULC", a, b) == _flat(M.f12_mul(_unflat(a), M.f12_conj(_unflat(b))))
    for P in (1, 2, 3):
        assert run("FROB%d" % P, a) == _flat(M.f12_frob(_unflat(a), P))
    c = M.final_exp_first_chunk(_unflat(a))
    assert run("EASY", a) == _flat(c)
    assert run("CYC", _flat(c)) == _flat(M.f12_cyclotomic_squared(c))
    c32 = c
    for _ in range(5): c32 = M.f12_cyclotomic_squared(c32)
    assert run("CYC5", _flat(c)) == _flat(c32)
    assert run("HARD", _flat(c)) == _flat(M.final_exp_last_chunk(c))
    assert run("FE", a) == _flat(M.final_exponentiation(_unflat(a)))
    # the Miller program: FE(machine Miller loop) == pairing of the oracle's model (groups/mod.rs:764-771), Jacobian inputs
    k1, k2 = rnd.randrange(1, M.R_ORD), rnd.randrange(1, M.R_ORD)
    P = M.g_mul(M.FQ_OPS, M.G1_ONE, k1); Qj = M.g_mul(M.FQ2_OPS, M.G2_ONE, k2)
    regs = gen.fresh_regs()
    regs[gen.IN_PX], regs[gen.IN_PY], regs[gen.IN_PZ] = (P[0], 0), (P[1], 0), (P[2], 0)
    regs[gen.IN_QX], regs[gen.IN_QY], regs[gen.IN_QZ] = Qj
    gen.run(B, progs["PAIRING"], regs)
    assert [regs[r] for r in gen.RES] == _flat(M.pairing(P, Qj))
    # the committed header is what the generator produces now
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()) as out:
        gen.emit(B, progs, ROOT / "bn_amd" / "csrc" / "wave_tables.hpp")
    assert "up to date" in out.getvalue(), "bn_amd/csrc/wave_tables.hpp is stale: run tools/gen_wave_tables.py"


def _rf12(oracle, rng):
    return np.concatenate([oracle.fp_from_int(FQ, int.from_bytes(rng.bytes(40), "little") % M.Q) for _ in range(12)])


def test_wave_machine_in_host_simulation(oracle, hs, kats):
    """wave.hpp's templates (the code the kernels instantiate) on a simulated wave, bounds enforced, against the oracle"""
    rng = np.random.default_rng(41)
    z = np.zeros(48, np.uint64)
    for _ in range(3):
        a, b = _rf12(oracle, rng), _rf12(oracle, rng)
        assert np.array_equal(hs.call("hsw_run", 0, a, b, out_words=96), oracle.fq12_mul(a, b))
        assert np.array_equal(hs.call("hsw_run", 1, a, b, out_words=96), oracle.fq12_mul(a, oracle.fq12_unitary_inverse(b)))
        for P in (1, 2, 3):
            assert np.array_equal(hs.call("hsw_run", 2 + P, a, z, out_words=96), oracle.fq12_frobenius_map(a, P))
        c = oracle.fq12_final_exp_first_chunk(a)
        assert np.array_equal(hs.call("hsw_run", 6, a, z, out_words=96), c)
        assert np.array_equal(hs.call("hsw_run", 2, c, z, out_words=96), oracle.fq12_cyclotomic_squared(c))
        c32 = c
        for _ in range(5): c32 = oracle.fq12_cyclotomic_squared(c32)
        assert np.array_equal(hs.call("hsw_run", 9, c, z, out_words=96), c32)                      # a fused run of five squarings
        assert np.array_equal(hs.call("hsw_run", 8, a, z, out_words=96), oracle.fq12_final_exponentiation(a))
    one = oracle.fq12_one()
    assert np.array_equal(hs.call("hsw_run", 8, one, z, out_words=96), one)
    # the reference's own vectors: fq12_test_vector start value (fields/mod.rs:83-169) through the product, and the cyclotomic
    # KAT's input (off the subgroup: only the full exponentiation is comparable) through the whole final exponentiation
    s = oracle.fq12_from_ints(kats["fq12_test_vector"]["start"])
    assert np.array_equal(hs.call("hsw_run", 0, s, s, out_words=96), oracle.fq12_sqr(s))
    t = oracle.fq12_from_ints(kats["test_cyclotomic_exp"]["orig"])
    assert np.array_equal(hs.call("hsw_run", 8, t, z, out_words=96), oracle.fq12_final_exponentiation(t))


def test_wave_pairing_in_host_simulation(oracle, hs, kats):
    """the whole pairing as ONE program of the wave machine (prologue, fused NAF Miller loop on the isomorphic curve, final
    exponentiation) on a simulated wave with bounds enforced: random Jacobian inputs, z = 1 inputs, infinity in either argument,
    and the reference's own known answer (groups/mod.rs:773-796, test_reduced_pairing)"""
    from bn_oracle import FR
    rng = np.random.default_rng(43)
    fr = lambda: oracle.fp_from_int(FR, int.from_bytes(rng.bytes(40), "little") % M.R_ORD)
    for _ in range(2):
        P = oracle.g1_mul(oracle.g1_one(), fr()); Q = oracle.g2_mul(oracle.g2_one(), fr())
        want = oracle.pairing(P, Q)
        assert np.array_equal(hs.call("hsw_pairing", 1, P, Q, out_words=96), want)
        m = hs.call("hsw_pairing", 0, P, Q, out_words=96)                       # Miller value: comparable after the exponentiation only
        assert np.array_equal(oracle.fq12_final_exponentiation(m), want)
    P, Q = oracle.g1_one(), oracle.g2_one()
    assert np.array_equal(hs.call("hsw_pairing", 1, P, Q, out_words=96), oracle.pairing(P, Q))
    one = oracle.fq12_one()
    assert np.array_equal(hs.call("hsw_pairing", 1, oracle.g1_zero(), Q, out_words=96), one)
    assert np.array_equal(hs.call("hsw_pairing", 1, P, oracle.g2_zero(), out_words=96), one)
    k1 = oracle.fp_from_decimal(FR, kats["test_reduced_pairing"]["k1"]); k2 = oracle.fp_from_decimal(FR, kats["test_reduced_pairing"]["k2"])
    gt = hs.call("hsw_pairing", 1, oracle.g1_mul(oracle.g1_one(), k1), oracle.g2_mul(oracle.g2_one(), k2), out_words=96)
    assert oracle.fq12_to_ints(gt) == [int(x) for x in kats["test_reduced_pairing"]["expected"]]

"""CPU tests of everything around the kernels: the C-ABI library loads and exports every declared symbol, the generated device
constants equal the reference's literals, sharding + the one all-gather of the multi-pairing (gloo, world_size 2), input
generation.  No GPU compute here."""
import json
import os
import pathlib
import re
import subprocess
import sys

import numpy as np
import pytest

import bn_model as M
from bn_oracle import FQ, FR

ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_c_abi_exports_every_declared_symbol():
    from bn_amd import _native
    hdr = (ROOT / "include" / "bn254_hip.h").read_text()
    declared = set(re.findall(r"^(?:int|void|size_t|const char \*|bn254_ctx \*)\s*\*?(bn254_\w+)\s*\(", hdr, re.M))
    assert len(declared) >= 45
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    _native.build()
    lib = _native.lib()                       # resolves every symbol, AttributeError otherwise
    for name in declared:
        assert hasattr(lib, name)
    assert lib.bn254_error_string(-1).decode().startswith("no usable HIP device")


def test_test_double_builds_and_stays_out_of_the_product():
    """tests/testdouble/: the one-lane-per-pairing kernels (a second GPU implementation for the parity tests) cross-compile for gfx950, export
    their five entry points - and the PRODUCT library holds none of them: no *_A kernel, no one-lane scalar-multiplication kernel"""
    import ctypes
    import testdouble
    lib = ctypes.CDLL(str(testdouble.build()))
    for name in ("bntd_miller", "bntd_final_exp", "bntd_gt_product", "bntd_g1_mul", "bntd_g2_mul"):
        assert hasattr(lib, name)
    sys.path.insert(0, str(ROOT / "tools"))
    import kernel_meta
    from bn_amd import _native
    product = set(kernel_meta.kernel_meta(_native.build()))
    assert not [k for k in product if k.endswith("_A") or k in ("bn254_g1_mul_k", "bn254_g2_mul_k")], product
    assert set(kernel_meta.kernel_meta(testdouble.LIB)) == {"bntd_miller_A", "bntd_final_exp_A", "bntd_gt_product_A", "bntd_g1_mul_k", "bntd_g2_mul_k"}
    # nothing under bn_amd/ refers to the test double or to the oracle
    for f in (ROOT / "bn_amd").rglob("*.py"):
        txt = f.read_text()
        assert "testdouble" not in txt and "bn_oracle" not in txt and "hostsim" not in txt, f


def test_fails_loudly_without_gpu():
    import bn_amd
    from bn_amd import _native
    if _native.lib().bn254_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_native.Bn254Error):
        bn_amd.Engine(0)
    with pytest.raises(_native.Bn254Error):
        bn_amd.pairing(bn_amd.G1.one(), bn_amd.G2.one())


def _hdr_arrays():
    txt = (ROOT / "bn_amd" / "csrc" / "bn254_constants.hpp").read_text()
    out = {}
    for m in re.finditer(r"BN254_CONSTANT uint32_t (\w+)((?:\[\d+\])+) = (\{.*?\});", txt, re.S):
        vals = [int(x, 16) for x in re.findall(r"0x[0-9a-f]+", m.group(3))]
        out[m.group(1)] = vals
    for m in re.finditer(r"BN254_CONSTANT uint(?:32|64)_t (\w+) = (0x[0-9a-f]+)", txt):
        out[m.group(1)] = int(m.group(2), 16)
    return out


def _val(limbs9):
    return sum(v << (29 * i) for i, v in enumerate(limbs9))


def test_device_constants_equal_reference_literals(ref_consts):
    """bn254_constants.hpp (radix 2^29, Montgomery 2^261, derived from u) -> reference image (radix 2^64, Montgomery 2^256)"""
    h = _hdr_arrays()
    R261 = 1 << 261
    def ref_image(limbs9):                       # internal Montgomery value -> the reference's 4 x u64 Montgomery limbs
        canon = _val(limbs9) * pow(R261, -1, M.Q) % M.Q
        return M.to_mont_limbs(canon)
    assert _val(h["Q"]) == M.Q and h["QINV"] == (-pow(M.Q, -1, 1 << 29)) % (1 << 29)
    assert _val(h["ONE"]) == R261 % M.Q and _val(h["C_IN"]) == (1 << 266) % M.Q and _val(h["C_OUT"]) == (1 << 256) % M.Q
    assert h["FE_MU"] == (1 << 285) // M.Q and h["FE_MU24"] == (1 << 277) // M.Q
    assert ref_image(h["TWO_INV"]) == ref_consts["two_inv"]
    assert [ref_image(h["G2_B"][:9]), ref_image(h["G2_B"][9:])] == ref_consts["g2_coeff_b"]
    b3 = [(3 * _val(h["G2_B"][:9])) % M.Q, (3 * _val(h["G2_B"][9:])) % M.Q]
    assert [_val(h["G2_3B"][:9]), _val(h["G2_3B"][9:])] == b3
    for name, key in (("FROB6_C1", "fq6_frobenius_coeffs_c1"), ("FROB6_C2", "fq6_frobenius_coeffs_c2"), ("FROB12_C1", "fq12_frobenius_coeffs_c1")):
        tab = h[name]
        assert len(tab) == 4 * 2 * 9
        for p in (1, 2, 3):
            c0 = tab[(2 * p) * 9:(2 * p + 1) * 9]; c1 = tab[(2 * p + 1) * 9:(2 * p + 2) * 9]
            lit = ref_consts[key][str(p)]
            assert ref_image(c0) == lit[0]
            assert (ref_image(c1) == lit[1]) if len(lit) == 2 else (_val(c1) == 0)
    assert [ref_image(h["TWIST_MUL_BY_Q_X"][:9]), ref_image(h["TWIST_MUL_BY_Q_X"][9:])] == ref_consts["twist_mul_by_q_x"]
    assert [ref_image(h["TWIST_MUL_BY_Q_Y"][:9]), ref_image(h["TWIST_MUL_BY_Q_Y"][9:])] == ref_consts["twist_mul_by_q_y"]
    assert [h["ATE_LOOP_LOW64"], 1, 0, 0] == ref_consts["ate_loop_count"]
    assert [h["BN_U"], 0, 0, 0] == ref_consts["exp_by_neg_z_u"]
    fr = h["FR_MOD32"]
    assert sum(v << (32 * i) for i, v in enumerate(fr)) == M.R_ORD and h["FR_INV32"] == ref_consts["Fr"]["inv"] & 0xffffffff


def test_shard_range_covers_everything():
    from bn_amd.distributed import shard_range
    for n in (0, 1, 7, 1 << 16, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1


def test_synthetic_scalars(oracle):
    from bn_amd import distributed as D
    a = D.synthetic_scalars(10, 20, 0); b = D.synthetic_scalars(10, 20, 1)
    assert a.shape == (10, 4) and not np.array_equal(a, b)
    assert np.array_equal(a[3:6], D.synthetic_scalars(13, 16, 0))            # index-addressed: shards agree with the whole
    for row in a:
        v = oracle.fp_to_int(FR, row)
        assert 0 <= v < M.R_ORD and np.array_equal(oracle.fp_from_int(FR, v), row)
    g1, g2 = D.generator_limbs()
    assert np.array_equal(g1, oracle.g1_one()) and np.array_equal(g2, oracle.g2_one())


WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path[:0] = [%(root)r, %(root)r + "/oracle", %(root)r + "/tests"]
import bn_oracle
from bn_amd import distributed as D
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
o = bn_oracle.Oracle()
class OracleEngine:                       # test double: same five methods as TorchEngine, CPU tensors, oracle arithmetic
    def _np(self, t): return t.numpy().view(np.uint64)
    def _t(self, a): return torch.from_numpy(np.ascontiguousarray(a).view(np.int64))
    def pairing_batch(self, p, q, out=None): return self._t(o.pairing_batch(self._np(p), self._np(q), 1))
    def miller_product(self, p, q):
        acc = o.fq12_one()
        for a, b in zip(self._np(p), self._np(q)): acc = o.fq12_mul(acc, o.miller_only(a, b))
        return self._t(acc)
    def gt_product(self, vals):
        acc = o.fq12_one()
        for v in self._np(vals): acc = o.fq12_mul(acc, v)
        return self._t(acc)
    def final_exp(self, f): return self._t(o.fq12_final_exponentiation(self._np(f)))
    def g2_prepare(self, q): return q                       # the double's "prepared handle" is the points themselves
    def miller_product_prepared(self, p, prepared): return self.miller_product(p, prepared)
n = %(n)d
data = np.load(%(data)r)
lo, hi = D.shard_range(n, rank, world)
eng = OracleEngine()
P = torch.from_numpy(data["P"][lo:hi].view(np.int64)); Q = torch.from_numpy(data["Q"][lo:hi].view(np.int64))
gt = D.pairing_product_sharded(eng, P, Q)
loc = D.pairing_batch_sharded(eng, P, Q)
gtp = D.pairing_product_prepared_sharded(eng, P, eng.g2_prepare(Q))          # the same exchange and tail over prepared points
assert torch.equal(gtp, gt)
np.save(%(out)r + f".{rank}.npy", np.concatenate([gt.numpy().view(np.uint64).reshape(1, 48), loc.numpy().view(np.uint64).reshape(-1, 48)]))
dist.barrier(); dist.destroy_process_group()
'''


def test_sharded_product_and_batch_gloo_world2(oracle, tmp_path):
    """the N>1 host path: contiguous shards, ONE all_gather of 384 B per rank, world-1 multiplications, one final exponentiation"""
    rng = np.random.default_rng(21)
    n = 7
    k1 = np.stack([oracle.fp_from_int(FR, int.from_bytes(rng.bytes(40), "little") % M.R_ORD) for _ in range(n)])
    k2 = np.stack([oracle.fp_from_int(FR, int.from_bytes(rng.bytes(40), "little") % M.R_ORD) for _ in range(n)])
    P = oracle.g1_mul_batch_jacobian(np.tile(oracle.g1_one(), (n, 1)), k1); Q = oracle.g2_mul_batch_jacobian(np.tile(oracle.g2_one(), (n, 1)), k2)
    P[2] = oracle.g1_zero()
    data = tmp_path / "pq.npz"; np.savez(data, P=P, Q=Q)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": str(ROOT), "n": n, "data": str(data), "out": str(tmp_path / "res")})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    assert [p.wait(timeout=300) for p in procs] == [0, 0]
    want = oracle.pairing_product(P, Q)
    allb = oracle.pairing_batch(P, Q)
    got = [np.load(str(tmp_path / f"res.{r}.npy")) for r in range(2)]
    assert np.array_equal(got[0][0], want) and np.array_equal(got[1][0], want)          # every rank holds the same Gt
    assert np.array_equal(np.concatenate([got[0][1:], got[1][1:]]), allb)                # shards concatenate to the batch


def test_cpp_facade_constants_match_oracle(oracle, tmp_path):
    """include/bn254.hpp mirrors src/lib.rs: its G1::one/G2::one/Gt::one/Fr::one images are the reference's"""
    from bn_amd import _native
    _native.build()
    src = tmp_path / "dump.cpp"
    src.write_text(r'''
#include "bn254.hpp"
#include <cstdio>
template <class T> void dump(const T &t) { const uint64_t *w = reinterpret_cast<const uint64_t *>(&t); for (size_t i = 0; i < sizeof(T) / 8; ++i) std::printf("%llu ", (unsigned long long)w[i]); std::printf("\n"); }
int main() { dump(bn::G1::one()); dump(bn::G1::zero()); dump(bn::G2::one()); dump(bn::G2::zero()); dump(bn::Gt::one()); dump(bn::Fr::one());
             return bn::G1::zero().is_zero() && !bn::G2::one().is_zero() ? 0 : 1; }
''')
    exe = tmp_path / "dump"
    subprocess.check_call(["g++", "-std=c++17", "-I", str(ROOT / "include"), str(src), "-o", str(exe),
                           "-L", str(ROOT / "bn_amd"), "-lbn254_hip", "-Wl,-rpath," + str(ROOT / "bn_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    lines = subprocess.check_output([str(exe)]).decode().strip().split("\n")
    got = [np.array([int(x) for x in l.split()], np.uint64) for l in lines]
    want = [oracle.g1_one(), oracle.g1_zero(), oracle.g2_one(), oracle.g2_zero(), oracle.fq12_one(), oracle.fp_from_int(FR, 1)]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_fr_from_str_matches_reference_semantics():
    """fields/fp.rs:39-59: ASCII decimal digits only (to_digit(10)); the empty string is zero; anything else is None"""
    import bn_amd
    assert bn_amd.Fr.from_str("") == bn_amd.Fr.zero()
    assert bn_amd.Fr.from_str("0012") == bn_amd.Fr(12)
    assert bn_amd.Fr.from_str(str(M.R_ORD + 5)) == bn_amd.Fr(5)            # reduced mod r like the reference's digit-by-digit Horner loop
    for bad in ("-1", "1 ", "0x10", "\u00b2", "\u0661\u0662", "1e3", "+7"):
        assert bn_amd.Fr.from_str(bad) is None, bad


def test_bench_relaunches_itself_for_multi_gpu():
    """`python bench.py --gpus 2` as a plain command (how the driver ran N = 1): with no rendezvous environment it must start 2
    ranks itself.  Without a GPU each rank stops at the "needs an MI355X" check - which proves both ranks were started."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu-marked bench test")
    assert out.returncode != 0
    # torchrun kills the other rank as soon as the first one exits, so only one of them is sure to print its message; that the
    # second rank existed is in torchrun's own failure report
    assert out.stderr.count("bench.py needs an MI355X") >= 1, out.stderr[-1500:]
    assert out.stderr.count("bench.py needs an MI355X") >= 2 or "local_rank: 1" in out.stderr, out.stderr[-1500:]


def test_scaling_prediction_is_made_of_measurements_only():
    """bench.py's expected_scaling: every number in the prediction comes from a one-GPU bench line (this run's, or the newest
    committed profiles/r*_bench_line.json that carries the four configs[3] shard sizes) - no literals: scaling the inputs scales the outputs, the
    committed source reproduces, and the round-4 constants are gone from the source"""
    sys.path.insert(0, str(ROOT))
    import bench
    src = (ROOT / "bench.py").read_text()
    body = src[src.index("def expected_scaling"):src.index("def bench_multi_c")]
    for stale in ("11.70", "6.47", "6.5,", "3.45", "2.18"):
        assert stale not in body, stale
    si = bench.scaling_inputs()
    assert si is not None and si["source"].startswith("profiles/r") and set(si["product_ms"]) == {"product_2_18", "product_2_17", "product_2_16", "product_2_15"}
    line = json.loads((ROOT / si["source"]).read_text().splitlines()[0])
    assert line["n_gpus"] == 1 and abs(si["round_ms"] - line["ms_per_step"]) < 1e-9
    for w, key in ((2, "product_2_17"), (4, "product_2_16"), (8, "product_2_15")):
        e = bench.expected_scaling("product", w, si)
        assert abs(e["ms_per_step"] - (line["side"][key]["ms_per_step"] + 0.05)) < 1e-9 and e["inputs_from"] == si["source"]
        assert bench.expected_scaling("pairing", w, si)["speedup_vs_1_gpu"] == float(w)
    twice = {"round_ms": 2 * si["round_ms"], "product_ms": {k: 2 * v for k, v in si["product_ms"].items()}, "source": "synthetic"}
    assert abs(bench.expected_scaling("pairing", 8, twice)["ms_per_step"] - 2 * bench.expected_scaling("pairing", 8, si)["ms_per_step"]) < 1e-9
    assert abs(bench.expected_scaling("product", 1, twice)["ms_per_step"] - 2 * bench.expected_scaling("product", 1, si)["ms_per_step"]) < 1e-9
    assert bench.expected_scaling("product", 4, None) is None                         # no measurements, no prediction
    # a line of THIS run that carries the side object takes precedence over the committed file
    fake = {"n_gpus": 1, "ms_per_step": 7.0, "config": {"pairings_per_gpu": 65536}, "side": {k: {"ms_per_step": 1.0} for k in si["product_ms"]}}
    assert bench.scaling_inputs(fake)["source"] == "this run" and bench.scaling_inputs(fake)["round_ms"] == 7.0

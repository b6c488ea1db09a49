"""GPU tests that guard the BUILD and the choice of kernel, not one operation:
  * the cross-path soak (tools/soak_paths.py): wave machine, four-lane and lane-pair kernels on the same pairings for 40 batch sizes
    around every threshold and wave / workgroup boundary, 2 % points at infinity, the multi-pairing through its three routes, a slice of
    every batch against the CPU oracle (pairing(p, q): /root/reference/src/groups/mod.rs:764-771);
  * a kernel unit REBUILT ON THE BOX with the library's own flags and run against the committed goldens: a compiler that starts folding
    DPP moves into arithmetic again (csrc/fe.hpp fe_lc4_core; round 4: wrong pairings from the four-lane Miller kernel) is caught by
    running the result, not only by the disassembly check of tests/test_build_quality.py."""
import os
import pathlib
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def te():
    import torch
    import bn_amd
    from bn_amd import distributed as D
    return D.TorchEngine(bn_amd.Engine(0), torch.device("cuda", 0))


def test_cross_path_soak(oracle, te):
    sys.path.insert(0, str(ROOT / "tools"))
    import soak_paths
    lines = []
    checked, bad = soak_paths.soak(te, oracle, log=lines.append)
    assert checked >= 450, checked            # 40 sizes x (three mappings + default + native prepared tables) + products (fused and over native tables, M = 1 / 2 / 4) + oracle slices
    assert bad == 0, [l for l in lines if "MISMATCH" in l]


_GOLDEN_RUNNER = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import bn_amd
from bn_amd import _native
assert str(_native.LIB_PATH) == sys.argv[2], _native.LIB_PATH
g = np.load(sys.argv[1] + "/tests/golden/pairing_goldens.npz")
e = bn_amd.Engine(0)
e.profile(True)
for opts, kernel in (({"wave_pairing_max": 0, "wave_fe_max": 0, "quad_max": 1 << 20}, "miller_quad"), ({"wave_pairing_max": 0, "wave_fe_max": 0, "quad_max": 0}, "miller"),
                     ({"wave_pairing_max": 1 << 20, "wave_fe_max": 1 << 20}, "pairing_wave")):
    with e.options(**opts):
        e.profile_reset()
        got = e.pairing_batch(g["g1"], g["g2"])
        assert e.kernel_stats(kernel)[1] >= 1, kernel
        assert np.array_equal(got, g["gt"]), "goldens differ through " + kernel
one1 = np.tile(g["g1"][0], (g["k1"].shape[0], 1))
assert np.array_equal(e.g1_mul_batch(one1, g["k1"]), g["g1"])
print("REBUILT-OK")
"""


def test_kernel_units_rebuilt_on_the_box_reproduce_the_goldens(tmp_path):
    """recompile the four-lane and the lane-pair kernel units HERE with the library's flags, link them with the shipped objects of the
    other units, and run the committed goldens through the four-lane, lane-pair and wave kernels of that library in a fresh process"""
    from bn_amd import _native
    hipcc = shutil.which("hipcc") or _native.HIPCC
    if not pathlib.Path(hipcc).exists():
        pytest.skip("no hipcc on this box")
    flags = (_native.OBJ_DIR / "flags.txt").read_text().split()
    assert "-amdgpu-dpp-combine=false" in flags, flags
    objs = []
    for src in _native.SOURCES:
        shipped = _native.OBJ_DIR / (src.stem + ".o")
        if src.stem in ("bn254_kernels_q", "bn254_kernels_b"):
            obj = tmp_path / (src.stem + ".o")
            objs.append(obj)
        else:
            assert shipped.exists(), f"{shipped} did not travel with the snapshot"
            objs.append(shipped)
    procs = [subprocess.Popen([hipcc] + flags[1:] + ["-c", str(src), "-o", str(tmp_path / (src.stem + ".o"))])
             for src in _native.SOURCES if src.stem in ("bn254_kernels_q", "bn254_kernels_b")]
    assert all(p.wait(timeout=900) == 0 for p in procs)
    lib = tmp_path / "libbn254_rebuilt.so"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [str(o) for o in objs] + ["-ldl", "-lpthread", "-o", str(lib)])
    # the rebuilt code holds no DPP instruction other than the plain move either
    sys.path.insert(0, str(ROOT / "tools"))
    import isa_mix
    import re
    bad = [l.strip()[:90] for text in isa_mix.disassemble(lib) for l in text.splitlines() if re.match(r"^\s+v_(?!mov_b32_dpp\s)\w+_dpp\s", l)]
    assert not bad, bad[:5]
    env = dict(os.environ, BN254_LIB_PATH=str(lib))
    r = subprocess.run([sys.executable, "-c", _GOLDEN_RUNNER, str(ROOT), str(lib)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "REBUILT-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_subrev_dpp_semantics_of_this_gpu(tmp_path):
    """Characterisation, not parity: WHY the library is built with LLVM's DPP combiner off (profiles/r05_dpp_fold_bisect.txt).  On the MI355X
    boxes of round 5 `v_subrev_u32_dpp d, x, y` computes dpp(y) - x - the lane permutation lands on the operand that becomes the minuend after
    the opcode's operand reversal - where LLVM's model (and the assembler syntax) say y - dpp(x); GCNDPPCombine emits exactly that instruction
    when a DPP value is the subtrahend, and one such fold made every four-lane pairing wrong in round 4.  The shipped library contains no DPP
    instruction other than v_mov_b32_dpp (tests/test_build_quality.py), so NEITHER behaviour affects the product; this test records which one
    the box under test has: the known quirk passes, the documented behaviour skips with a note (the guard could be revisited there), anything
    else fails (an unknown third behaviour deserves a look before trusting DPP at all)."""
    from bn_amd import _native
    hipcc = shutil.which("hipcc") or _native.HIPCC
    if not pathlib.Path(hipcc).exists():
        pytest.skip("no hipcc on this box")
    exe = tmp_path / "dpp_subrev_check"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", str(ROOT / "tools" / "variants" / "dpp_subrev_check.hip"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120).stdout
    import re
    rows = {}
    for line in out.splitlines():
        m = re.match(r"(v_\w+)\s+d,.*?(\[[\d,]+\]|\(no DPP\)).*?y - dpp\(x\)\s+(\d+) \| dpp\(x\) - y\s+(\d+) \| dpp\(y\) - x\s+(\d+)", line)
        if m:
            rows[(m.group(1), m.group(2))] = tuple(int(x) for x in m.group(3, 4, 5))
    assert ("v_subrev_u32_dpp", "[1,0,3,2]") in rows and ("v_sub_u32_dpp", "[1,0,3,2]") in rows, out
    assert rows[("v_sub_u32_dpp", "[1,0,3,2]")][1] == 64, out                    # the plain form: dpp(x) - y, as documented
    doc, _, quirk = rows[("v_subrev_u32_dpp", "[1,0,3,2]")]
    if doc == 64:
        pytest.skip("v_subrev_u32_dpp follows the documented operand order on this box (y - dpp(x)): the round-4 miscompile would not occur here")
    assert quirk == 64, out

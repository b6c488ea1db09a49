"""Static checks on the gfx950 code the build produced (no GPU needed: the code objects are disassembled from libbn254_hip.so).

The pairing kernels are register-allocated to the last VGPR, across out-of-line calls as well: an edit to a COLD helper can move the
allocation of the hot loops (round 3: a cheaper Frobenius map in the final exponentiation put 15-22 scratch accesses into its squaring
and product blocks and cost 16 % - profiles/r03y_ab_frobenius.txt).  This keeps such a regression from passing unnoticed."""
import collections
import pathlib
import re
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))


def _blocks(text, pat):
    """(instructions, scratch accesses) of every basic block of the functions whose name contains `pat`"""
    out = []; on = False; cur = collections.Counter()
    def flush():
        nonlocal cur
        if sum(cur.values()): out.append((sum(cur.values()), cur["scratch"]))
        cur = collections.Counter()
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            if on: flush()
            on = pat in m.group(1); continue
        if not on: continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if not m: continue
        op = m.group(1)
        cur["scratch" if op.startswith("scratch_") else "other"] += 1
        if op.startswith(("s_cbranch", "s_branch")): flush()
    if on: flush()
    return out


@pytest.mark.parametrize("kernel", ["bn254_miller_naf_B", "bn254_final_exp_B"])
def test_hot_loops_are_spill_free(kernel):
    import isa_mix
    so = ROOT / "bn_amd" / "libbn254_hip.so"
    if not so.exists() or not (isa_mix.LLVM / "llvm-objdump").exists():
        pytest.skip("library or llvm-objdump not present")
    blocks = [b for text in isa_mix.disassemble(so) for b in _blocks(text, kernel)]
    hot = [b for b in blocks if b[0] >= 2500]                  # the loop bodies: squarings, line / table products (2.8k .. 14k instructions)
    assert len(hot) >= 2, blocks
    assert all(s == 0 for _, s in hot), f"scratch accesses inside the hot blocks of {kernel}: {hot}"

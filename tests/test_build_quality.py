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


# bn254_gt_pow_B: its 13 spilled VGPRs (round 3: 17; the table construction is out of line and the scalar is fetched after it since round 4)
# are loop-invariant per-lane scalars - pair index, the input / output / table addresses - stored in the prologue and reloaded in the
# epilogue and between the blocks (llvm's "Folded Spill" / "Folded Reload" annotations); what matters is checked here
@pytest.mark.parametrize("kernel", ["bn254_miller_naf_B", "bn254_miller_native_B", "bn254_final_exp_B", "bn254_gt_pow_B", "bn254_miller_naf_Q", "bn254_final_exp_Q"])
def test_hot_loops_are_spill_free(kernel):
    import isa_mix
    so = ROOT / "bn_amd" / "libbn254_hip.so"
    if not so.exists() or not (isa_mix.LLVM / "llvm-objdump").exists():
        pytest.skip("library or llvm-objdump not present")
    blocks = [b for text in isa_mix.disassemble(so) for b in _blocks(text, kernel)]
    # the loop bodies: squarings, line / table products (2.7k .. 13k instructions; on four lanes a Granger-Scott squaring is 1.9k)
    hot = [b for b in blocks if b[0] >= (1500 if kernel.endswith("_Q") else 2500)]
    # (the native prepared Miller loop is ONE run of 8.6k instructions between branches: squaring and line product, the addition steps enter it in the middle)
    assert len(hot) >= (1 if kernel == "bn254_miller_native_B" else 2), blocks
    if kernel == "bn254_miller_naf_B":
        hot = hot[1:]               # the first big block is the prologue (both affine conversions around the inversion call), run once
    if kernel == "bn254_gt_pow_B":
        hot = hot[-3:]              # the window loop: two squaring bodies and the table product (the product that BUILDS the table comes first and may spill)
    assert all(s == 0 for _, s in hot), f"scratch accesses inside the hot blocks of {kernel}: {hot}"


# vgpr_spill_count ceilings of EVERY shipped kernel (tools/kernel_meta.py reads them from the code objects).  The ceilings are the
# values of the build this test was written against: anything above means an edit moved a register allocation - look at
# `tools/isa_mix.py --blocks KERNEL` before raising one.  (The one-lane test double lives in tests/testdouble/, not in this library.)
SPILL_CEILING = {
    "bn254_miller_B": 3, "bn254_miller_naf_B": 0, "bn254_final_exp_B": 7, "bn254_miller_shared2_B": 19, "bn254_miller_shared4_B": 19,
    "bn254_g2_precompute_B": 0, "bn254_miller_prepared_B": 0, "bn254_g2_prepare_native_B": 0, "bn254_miller_native_B": 0, "bn254_miller_native_shared2_B": 0, "bn254_miller_native_shared4_B": 0, "bn254_native_identity_B": 0, "bn254_gt_mul_B": 0, "bn254_gt_pow_B": 11, "bn254_gt_inverse_B": 4,
    "bn254_exp_by_neg_z_B": 4, "bn254_miller_naf_Q": 0, "bn254_final_exp_Q": 0,
    "bn254_g1_mul_M": 7, "bn254_g1_mul_chain_M": 0,      # (round 6: 7, all in the loop's preheader - the digit streams and the GLV halves are set up there -
                                                         #  and the epilogue; the window loop itself: test_scalar_multiplication_loops_do_not_store_to_scratch)
    "bn254_g2_mul_M": 0, "bn254_g2_mul_chain_M": 0, "bn254_g1_add_M": 0, "bn254_g2_add_M": 0,
    "bn254_final_exp_W": 0, "bn254_pairing_W": 0, "bn254_gt_tail_W": 0, "bn254_wave_ubench_W": 0, "bn254_gt_reduce_W": 2,
    "bn254_g1_encode_k": 0, "bn254_g2_encode_k": 0, "bn254_g1_decode_k": 0, "bn254_g2_decode_k": 18, "bn254_fr_encode_k": 0, "bn254_fr_decode_k": 0,
    "bn254_ubench_mad_k": 0, "bn254_synthetic_scalars_k": 0, "bn254_tile_k": 0,
}
UNGUARDED = set()
# kernels that must fit their occupancy target without private memory beyond small call frames: the hot state of the scalar
# multiplications used to be written to scratch on every addition (round 4: 10.7 KB per G1 multiplication) - private memory that
# is only the window-table setup stays below these sizes
PRIVATE_CEILING = {"bn254_g1_mul_M": 1400, "bn254_g2_mul_M": 1400, "bn254_miller_naf_B": 160, "bn254_miller_B": 176, "bn254_miller_native_B": 0}


def test_spill_ceilings_of_every_kernel():
    import isa_mix
    import kernel_meta
    so = ROOT / "bn_amd" / "libbn254_hip.so"
    if not so.exists() or not (isa_mix.LLVM / "llvm-readelf").exists():
        pytest.skip("library or llvm-readelf not present")
    meta = kernel_meta.kernel_meta(so)
    unknown = set(meta) - set(SPILL_CEILING) - UNGUARDED
    assert not unknown, f"kernels without a spill ceiling (add them to SPILL_CEILING): {sorted(unknown)}"
    missing = set(SPILL_CEILING) - set(meta)
    assert not missing, f"kernels named in SPILL_CEILING that the library no longer has: {sorted(missing)}"
    over = {k: (meta[k]["spill"], c) for k, c in SPILL_CEILING.items() if meta[k]["spill"] > c}
    assert not over, f"vgpr_spill_count above its ceiling (kernel: (now, ceiling)): {over}"
    over = {k: (meta[k]["private"], c) for k, c in PRIVATE_CEILING.items() if meta[k]["private"] > c}
    assert not over, f"private segment above its ceiling (kernel: (now, ceiling)): {over}"


def test_scalar_multiplication_loops_do_not_store_to_scratch():
    """the window loop of the G1 / G2 scalar multiplications (doublings + mixed additions: blocks of >= 900 instructions with
    multiply-adds) holds the running point in registers: no scratch STORES (round 4: a by-reference cold call made the compiler
    store the point after every addition)"""
    import isa_mix
    so = ROOT / "bn_amd" / "libbn254_hip.so"
    if not so.exists() or not (isa_mix.LLVM / "llvm-objdump").exists():
        pytest.skip("library or llvm-objdump not present")
    for kernel in ("bn254_g1_mul_M", "bn254_g2_mul_M"):
        blocks = []                                   # (instructions, scratch stores) of every basic block, in address order
        for text in isa_mix.disassemble(so):
            on = False; n = st = 0
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    if on and n: blocks.append((n, st))
                    on = kernel + "E" in m.group(1) or m.group(1).endswith(kernel); n = st = 0; continue
                if not on: continue
                m = re.match(r"^\s+([a-z_0-9]+)\s", line)
                if not m: continue
                op = m.group(1); n += 1
                if op.startswith("scratch_store"): st += 1
                if op.startswith(("s_cbranch", "s_branch")):
                    blocks.append((n, st)); n = st = 0
            if on and n: blocks.append((n, st))
        big = [i for i, (n, _) in enumerate(blocks) if n >= 900]
        assert len(big) >= 4, (kernel, blocks)
        # from the doubling block of the window loop (the fourth big block from the end: doubling, two halves of the mixed addition,
        # the normalisation that follows the loop) to the end of the function, small glue blocks included - the table construction
        # before it may store its Jacobian table
        region = blocks[big[-4]:]
        assert sum(st for _, st in region) == 0, f"scratch stores inside the window loop of {kernel}: {region}"


def test_only_plain_dpp_moves_and_the_flag_that_guarantees_them():
    """Round 4: when the quotient estimate of the fused reductions read a neighbour pair's top limb directly, LLVM's DPP combiner folded
    the quad_perm move into the subtraction (v_sub_u32_dpp / v_subrev_u32_dpp) and the four-lane Miller kernel returned wrong values for
    every pairing on the GPU - right again with -mllvm -amdgpu-dpp-combine=false.  Round 5 found the cause (profiles/r05_dpp_fold_bisect.txt):
    on gfx950 `v_subrev_u32_dpp d, x, a` computes dpp(a) - x, not a - dpp(x) as LLVM models it.  Since round 5 EVERY unit is built with the combiner off
    (bn_amd/_native.py DEVICE_FLAGS; free: profiles/r05_ab_dpp_combine_off.txt) and this test - which does not skip: it builds the
    library if it has to - rejects ANY DPP instruction other than the plain v_mov_b32_dpp in the shipped code objects (round 4's regex
    let the same combiner's 34 v_add_u32_dpp through).  tests/test_gpu_soak.py re-runs the goldens on kernel units rebuilt on the GPU box."""
    import isa_mix
    from bn_amd import _native
    so = _native.build()
    assert (isa_mix.LLVM / "llvm-objdump").exists() and (isa_mix.LLVM / "clang-offload-bundler").exists(), "the ROCm LLVM tools are part of the image"
    assert "-amdgpu-dpp-combine=false" in _native.DEVICE_FLAGS and "-amdgpu-dpp-combine=false" in (_native.OBJ_DIR / "flags.txt").read_text()
    ops = collections.Counter(m.group(1) for text in isa_mix.disassemble(so) for m in re.finditer(r"^\s+(v_\w+_dpp)\s", text, re.M))
    assert ops["v_mov_b32_dpp"] > 10000, ops                       # the lane-pair exchanges are there ...
    assert set(ops) == {"v_mov_b32_dpp"}, ops                       # ... and nothing else carries a DPP modifier


def test_the_measurement_switch_builds(tmp_path):
    """The library has three compile-time switches left (round 4: 45): BN_INLINE_ALL (every kernel unit defines it), BN_MILLER_HOOK / BN_EXP_HOOK / BN_MUL_HOOK / BN_G1_HOOK (the
    hooks of the hand-over policy of bn254_kernels_b.hip resp. of the G2 kernel of bn254_kernels_mul.hip: defined there, each with its own default) and BN_AB_ALIAS_SCRATCH - the zero-traffic twin of the table-carrying kernels
    (same instruction stream on a cache-resident footprint, WRONG results: a timing experiment only, profiles/r04a_ab_traffic_cost.txt,
    r05_ab_shared_miller_state_traffic.txt).  The first two are exercised by every build; this test builds the third in both units that
    know it, in parallel, device code only, and checks that nothing else in csrc/ is switchable."""
    import subprocess
    from bn_amd import _native
    csrc = ROOT / "bn_amd" / "csrc"
    switches = set()
    for f in list(csrc.glob("*.hpp")) + list(csrc.glob("*.hip")):
        if f.name in ("fe_asm.hpp", "wave_tables.hpp", "bn254_constants.hpp"):
            continue
        switches |= set(re.findall(r"^#\s*if(?:n?def|\s+!?defined\()\s*(BN_\w+)", f.read_text(), re.M))
    assert switches == {"BN_HOSTSIM", "BN_BOUNDS", "BN_INLINE_ALL", "BN_MILLER_HOOK", "BN_EXP_HOOK", "BN_MUL_HOOK", "BN_G1_HOOK", "BN_AB_ALIAS_SCRATCH"}, switches
    procs = [subprocess.Popen([_native.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value"] + _native.DEVICE_FLAGS +
                              ["-DBN_AB_ALIAS_SCRATCH", "--cuda-device-only", "-c", str(csrc / (u + ".hip")), "-o", str(tmp_path / (u + ".o"))],
                              stderr=subprocess.PIPE, text=True) for u in ("bn254_kernels_mul", "bn254_kernels_b")]
    for pr in procs:
        err = pr.communicate(timeout=1200)[1]
        assert pr.returncode == 0, err[-2000:]

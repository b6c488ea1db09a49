#!/usr/bin/env python3
"""ON THE GPU BOX, once per library (BN254_LIB_PATH): do the four-lane kernels of this library reproduce the committed goldens?
A: pairing_batch forced onto the four-lane Miller + four-lane final exponentiation; B: the four-lane MILLER values only (multi-pairing product:
four-lane Miller loops -> product tree -> ONE wave-cooperative final exponentiation) against the product of the golden pairings."""
import os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
import bn_amd
g = np.load(ROOT / "tests/golden/pairing_goldens.npz")
e = bn_amd.Engine(0)
with e.options(wave_pairing_max=0, wave_fe_max=0, quad_max=1 << 20):
    e.profile(True); e.profile_reset()
    got = e.pairing_batch(g["g1"], g["g2"])
    assert e.kernel_stats("miller_quad")[1] == 1 and e.kernel_stats("final_exp_quad")[1] == 1
    a_ok = bool(np.array_equal(got, g["gt"])); nbad = int((got != g["gt"]).any(axis=1).sum())
    prod = e.pairing_product(g["g1"], g["g2"])
with e.options(wave_pairing_max=0, wave_fe_max=0, quad_max=0):
    ref = e.pairing_product(g["g1"], g["g2"])                   # lane-pair Miller loops, same tail
    ok_ref = bool(np.array_equal(e.pairing_batch(g["g1"], g["g2"]), g["gt"]))
print(os.path.basename(os.environ.get("BN254_LIB_PATH", "default")), "A four-lane pairing == goldens:", a_ok, f"({nbad} of {len(got)} differ)",
      "| B four-lane Miller -> product == lane-pair Miller -> product:", bool(np.array_equal(prod, ref)), "| lane-pair pairing == goldens:", ok_ref)

#!/usr/bin/env python3
"""Generates tests/golden/pairing_goldens.npz (SURVEY.md section 8c "fixtures to commit") with the CPU oracle, AFTER the oracle
has passed every known-answer vector of the reference (tests/test_oracle_kats.py).  Data only: scalars in, limbs out.

  k1, k2        (N,4) u64   Fr scalars (Montgomery limbs), edge cases first: 1, 2, r-1, r-2, small, sparse, dense, then seeded random
  gt            (N,48) u64  pairing(k1*G1, k2*G2)              (bn::pairing, lib.rs:181)
  g1, g2        the normalized points k1*G1, k2*G2             (lib.rs:88-95)
  inf_*         pairs with a point at infinity -> Gt::one()    (groups/mod.rs:766)
  coeffs        (102,24) u64 prepared line coefficients of g2[5]   (groups/mod.rs:557-588)
  wire_g1/g2    encoded records of the first 32 points         (groups/mod.rs:143-160)
usage: python tests/golden/make_goldens.py      (from the repository root)
"""
import pathlib, sys
import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import bn_oracle  # noqa: E402

FR = 1
R_ORD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def main():
    bn_oracle.build()
    o = bn_oracle.Oracle()
    rng = np.random.default_rng(0x424E323534)
    edge = [1, 2, R_ORD - 1, R_ORD - 2, 3, 65537, 1 << 128, (1 << 253) + 1, (1 << 200) - 1, R_ORD // 2, R_ORD // 3]
    n = 96
    s1 = edge + [int.from_bytes(rng.bytes(40), "little") % R_ORD for _ in range(n - len(edge))]
    s2 = edge[::-1] + [int.from_bytes(rng.bytes(40), "little") % R_ORD for _ in range(n - len(edge))]
    k1 = np.stack([o.fp_from_int(FR, s) for s in s1]); k2 = np.stack([o.fp_from_int(FR, s) for s in s2])
    g1 = np.stack([o.g1_normalize(o.g1_mul(o.g1_one(), k)) for k in k1])
    g2 = np.stack([o.g2_normalize(o.g2_mul(o.g2_one(), k)) for k in k2])
    gt = o.pairing_batch(g1, g2)
    coeffs = o.g2_precompute(g2[5][:16])             # normalized point: z == 1, affine = (x, y)
    wire_g1 = np.stack([o.g1_encode(p) for p in g1[:32]]); wire_g2 = np.stack([o.g2_encode(p) for p in g2[:32]])
    np.savez(ROOT / "tests/golden/pairing_goldens.npz", k1=k1, k2=k2, g1=g1, g2=g2, gt=gt, coeffs=np.asarray(coeffs).reshape(102, 24),
             wire_g1=wire_g1, wire_g2=wire_g2, gt_one=o.fq12_one(), scalars1=np.array([str(s) for s in s1]), scalars2=np.array([str(s) for s in s2]))
    print("wrote", ROOT / "tests/golden/pairing_goldens.npz")


if __name__ == "__main__":
    main()

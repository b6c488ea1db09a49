"""bn_amd - MI355X-native batched BN254 optimal-ate pairing engine (HIP/gfx950) behind the API surface of the
reference crate zcash-hackworks/bn (`Fr`, `G1`, `G2`, `Gt`, `pairing`; src/lib.rs).

Everything computes on the GPU through libbn254_hip.so (C ABI: include/bn254_hip.h).  There is no CPU fallback: importing is
cheap, but any computation without the built library and a HIP device raises.
"""
from .engine import Engine, MultiEngine, PreparedG2 as PreparedG2Handle, FR_BYTES, G1_WORDS, G2_WORDS, GT_WORDS  # noqa: F401
from .api import Fr, G1, G2, Gt, PreparedG2, pairing, pairing_batch, pairing_product  # noqa: F401

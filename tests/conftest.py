import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import bn_oracle
    bn_oracle.build()
    return bn_oracle.Oracle()


@pytest.fixture(scope="session")
def kats():
    import json
    return json.loads((ROOT / "tests/golden/reference_kats.json").read_text())


@pytest.fixture(scope="session")
def ref_consts():
    import json
    return json.loads((ROOT / "tests/golden/reference_consts.json").read_text())


def canon_infinity(points):
    """parity definition for group outputs: the affine image (x, y, 1), or G::zero() = (0, 1, 0) for
    infinity.  The reference's normalize() leaves an infinite point's x, y as its addition chain happened to produce them
    (e.g. doubling (0,1,0) gives (0,-8,0)); those bytes carry no meaning, so both sides are mapped to (0, 1, 0)."""
    import numpy as np
    pts = np.array(points, dtype=np.uint64, copy=True)
    flat = pts.reshape(-1, pts.shape[-1])
    w = flat.shape[1] // 3
    one = np.array([0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f], np.uint64)
    for row in flat:
        if not row[2 * w:].any():
            row[:] = 0
            row[w:w + 4] = one
    return pts

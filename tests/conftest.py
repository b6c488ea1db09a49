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

import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "pim-jpeg-decoder_amd", "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def port():
    import oracle_lib
    return oracle_lib.Port()


@pytest.fixture(scope="session")
def ref():
    import oracle_lib
    if not oracle_lib.Ref.available():
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt copy)")
    return oracle_lib.Ref()


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(HERE, "golden", "manifest.json")) as f:
        return json.load(f)


def golden_bytes(name):
    with open(os.path.join(HERE, "golden", name + ".jpg"), "rb") as f:
        return f.read()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REF_XML = "/root/reference/output/DexterousDynamos.xml"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hand_model():
    """The north-star model, from the committed compiled fixture (the GPU box has no
    /root/reference); tests/golden/make_golden.py regenerates it from the reference XML+STLs."""
    import mujoco_amd
    return mujoco_amd.Model.load(os.path.join(GOLDEN, "dexterous_model.npz"))


@pytest.fixture(scope="session")
def have_reference():
    return os.path.exists(REF_XML)

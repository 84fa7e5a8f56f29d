import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.bindings import Oracle
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.bindings import Ref, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference; run `make -C oracle ref`)")
    return Ref()


@pytest.fixture(scope="session")
def lib():
    from grtcode_amd import api
    return api.load_library()     # raises LibraryMissing: no fallback


@pytest.fixture(scope="session")
def device(lib):
    from grtcode_amd import api
    return api.create_device(0)   # GRTCODE_GPU_ERR without a GPU: gpu tests fail loudly

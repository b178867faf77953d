import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle_lib import Ref, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libalacref.so not built (needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    import alac_amd
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return alac_amd.Context(0)

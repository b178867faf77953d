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
def gpu_ctx(request):
    """One context per test session.  tests/test_gpu_variants.py runs the parity files again as nested in-process sessions
    whose config carries `alac_variant`: the code-path options (alac_hip_set_option) that session's context is pinned to."""
    import torch
    import alac_amd
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    ctx = alac_amd.Context(0)
    for k, v in getattr(request.config, "alac_variant", {}).items():
        ctx.set_option(k, v)
    return ctx

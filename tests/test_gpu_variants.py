"""The alternative code paths behind the same C-ABI must pass the same parity tests: the first-generation
lane-per-chain encoder / decoder (options encoder_lane / decoder_lane), the un-fused launches (fused = 0, dec_fused = 0),
and the regimes the batch size normally selects: the small batches of the parity files run on the
four-lanes-per-chain mapping by default, so narrow = 0 puts them on the two-lane kernels of the 10 000-packet benchmark
and thru = 1 on the throughput regime's launches (class compaction, 8-taps-in-a-lane search, lazy word stores; DESIGN.md
4.0).  The switches are per-context options (alac_hip_set_option), so every variant runs IN THIS PROCESS: a nested pytest
session over the parity files whose `gpu_ctx` fixture (tests/conftest.py) creates a context pinned to the variant."""
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["tests/test_gpu_encode.py", "tests/test_gpu_decode.py", "tests/test_gpu_foreign.py", "tests/test_gpu_fuzz.py",
         "tests/test_gpu_multichannel.py", "tests/test_gpu_wholefile.py", "tests/test_gpu_chained.py"]

VARIANTS = {
    "first-generation": {"encoder_lane": 1, "decoder_lane": 1},
    "stagewise": {"fused": 0, "dec_fused": 0, "dec_direct": 2},  # direct reads at every batch size (automatic: from 80 000 packets)
    "two-lane-latency-regime": {"narrow": 0},
    "two-lane-latency-regime-unfolded": {"narrow": 0, "fold": 0},
    "throughput-regime": {"thru": 1},
    "positions-not-overlapped": {"overlap_pos": 0},
    "unfused-decode-chains-not-paired": {"dec_fused": 0, "dec_pair": 0},
    "unfused-decode-from-a-staged-copy": {"dec_fused": 0, "dec_direct": 0},
    "tiny-batch-coder-not-split": {"split_coder": 0},
}


class _Variant:
    """plugin of the nested session: hands the options to conftest's gpu_ctx and counts what ran"""

    def __init__(self, opts):
        self.opts = opts
        self.passed = self.failed = 0
        self.reports = []

    def pytest_configure(self, config):
        config.alac_variant = self.opts

    def pytest_runtest_logreport(self, report):
        if report.when == "call" and report.passed:
            self.passed += 1
        if report.failed:
            self.failed += 1
            self.reports.append(f"{report.nodeid}: {report.longreprtext[-1500:]}")


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant_passes_the_parity_files(gpu_ctx, name):
    plug = _Variant(VARIANTS[name])
    files = [os.path.join(ROOT, f) for f in FILES if os.path.exists(os.path.join(ROOT, f))]
    rc = pytest.main(["-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "--no-header", "-W", "ignore::pytest.PytestAssertRewriteWarning"] + files,
                     plugins=[plug])
    assert rc == 0 and plug.failed == 0, "\n".join(plug.reports)
    assert plug.passed > 50  # the files really ran (a collection error would also give rc != 0)

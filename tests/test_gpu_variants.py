"""The alternative code paths behind the same C-ABI must pass the same parity tests: the first-generation
lane-per-chain encoder / decoder (ALAC_HIP_ENCODER=lane, ALAC_HIP_DECODER=lane) and the un-fused launches
(ALAC_HIP_FUSED=0, ALAC_HIP_DEC_FUSED=0, ALAC_HIP_IDLEFAST=0), and the regimes the batch size normally selects: the small
batches of the parity files run on the four-lanes-per-chain mapping by default, so ALAC_HIP_NARROW=0 puts them on the
two-lane kernels of the 10 000-packet benchmark and ALAC_HIP_THRU=1 on the throughput regime's separate launches, class
compaction, 8-taps-in-a-lane search and lazy word stores.  The switches are read once per process, so each
variant runs the parity files in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["tests/test_gpu_encode.py", "tests/test_gpu_decode.py", "tests/test_gpu_fuzz.py", "tests/test_gpu_multichannel.py",
         "tests/test_gpu_wholefile.py"]


@pytest.mark.parametrize("env", [
    {"ALAC_HIP_ENCODER": "lane", "ALAC_HIP_DECODER": "lane"},
    {"ALAC_HIP_FUSED": "0", "ALAC_HIP_DEC_FUSED": "0"},
    {"ALAC_HIP_IDLEFAST": "0"},
    {"ALAC_HIP_PUBFENCE": "1"},
    {"ALAC_HIP_NARROW": "0"},
    {"ALAC_HIP_THRU": "1"},
    {"ALAC_HIP_THRU": "1", "ALAC_HIP_SUBBATCH": "2"},
    {"ALAC_HIP_OVERLAP_POS": "0"},
    {"ALAC_HIP_DEC_FUSED": "0", "ALAC_HIP_DEC_WIDE": "0"},
    {"ALAC_HIP_SPLIT_CODER": "0"},
], ids=["first-generation", "unfused", "idle-checked", "release-fence", "two-lane-latency-regime", "throughput-regime",
        "throughput-regime-sub-batches", "positions-not-overlapped", "unfused-two-lane-decode-predictor",
        "tiny-batch-coder-not-split"])
def test_variant_passes_the_parity_files(gpu_ctx, env):
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + FILES,
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]

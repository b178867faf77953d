"""The alternative code paths behind the same C-ABI must pass the same parity tests: the first-generation
lane-per-chain encoder / decoder (ALAC_HIP_ENCODER=lane, ALAC_HIP_DECODER=lane) and the un-fused launches
(ALAC_HIP_FUSED=0, ALAC_HIP_DEC_FUSED=0, ALAC_HIP_IDLEFAST=0).  The switches are read once per process, so each
variant runs the parity files in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["tests/test_gpu_encode.py", "tests/test_gpu_decode.py", "tests/test_gpu_fuzz.py", "tests/test_gpu_multichannel.py"]


@pytest.mark.parametrize("env", [
    {"ALAC_HIP_ENCODER": "lane", "ALAC_HIP_DECODER": "lane"},
    {"ALAC_HIP_FUSED": "0", "ALAC_HIP_DEC_FUSED": "0"},
    {"ALAC_HIP_IDLEFAST": "0"},
    {"ALAC_HIP_PUBFENCE": "1"},
], ids=["first-generation", "unfused", "idle-checked", "release-fence"])
def test_variant_passes_the_parity_files(gpu_ctx, env):
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + FILES,
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]

"""The device build of the synthetic PCM generator (alac_hip_synth_pcm, alac_synth.hip) writes the same bytes as the
host build (both compile alac_synth_core.h) and both match the pinned hashes of tests/golden/known_answers.json."""
import json
import os

import numpy as np
import pytest

import alac_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("depth,channels", [(16, 2), (24, 2), (20, 2), (32, 2), (16, 1), (24, 1)])
def test_device_generator_matches_host_and_pins(gpu_ctx, oracle, depth, channels):
    with open(os.path.join(ROOT, "tests", "golden", "known_answers.json")) as f:
        known = json.load(f)["synthetic"][f"{depth}bit_{channels}ch"]
    fmt = alac_amd.make_format(4096, depth, channels)
    n = known["frames"]
    dev = gpu_ctx.synth_pcm(0, n, fmt)
    gpu_ctx.synchronize()
    dev = dev.cpu().numpy()
    assert np.array_equal(dev, alac_amd.synth_pcm(0, n, fmt))
    assert f"{oracle.fnv(dev):016x}" == known["pcm_fnv"]


def test_device_generator_far_frames_and_odd_sizes(gpu_ctx):
    # shard edges of BASELINE configs[3]: frame indices around k * 125 000, and a frame size that is not a multiple of 4
    for first, n, frame in ((124_990, 20, 4096), (999_990, 10, 4096), (7, 13, 1023)):
        fmt = alac_amd.make_format(frame, 16, 2)
        dev = gpu_ctx.synth_pcm(first, n, fmt)
        gpu_ctx.synchronize()
        assert np.array_equal(dev.cpu().numpy(), alac_amd.synth_pcm(first, n, fmt))

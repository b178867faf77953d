"""GPU parity: HIP encode path vs the CPU oracle, through the C-ABI.  Bit-exact (integer/byte work)."""
import numpy as np
import pytest

import alac_amd

pytestmark = pytest.mark.gpu


def _oracle_stream(oracle, fmt, pcm, total_samples, segment_packets):
    enc = oracle.encoder(fmt.frame_size, fmt.bit_depth, fmt.num_channels, fmt.sample_rate)
    return enc.encode_stream(pcm, total_samples, segment_packets)


@pytest.mark.parametrize("depth,channels", [(16, 2), (24, 2), (20, 2), (32, 2), (16, 1), (24, 1), (20, 1), (32, 1)])
def test_independent_packets_match_oracle(gpu_ctx, oracle, depth, channels):
    import torch
    fmt = alac_amd.make_format(4096, depth, channels)
    n = 48
    pcm = alac_amd.synth_pcm(0, n, fmt)
    stream, sizes = gpu_ctx.encode_to_host(fmt, torch.from_numpy(pcm).cuda(), n)
    ref, ref_sizes = _oracle_stream(oracle, fmt, pcm, n * 4096, 1)
    assert np.array_equal(sizes, ref_sizes)
    assert np.array_equal(stream, ref)

"""alac_amd — MI355X-native ALAC encode/decode hot path.

The product is libalac_hip.so (HIP kernels for gfx950 behind the C-ABI of include/alac_hip.h,
plus the C++ ALACEncoder/ALACDecoder classes).  This package is the thin Python binding used by
the tests and bench.py; it never falls back to a CPU implementation.
"""
from .capi import (AlacError, Comm, Context, Format, LIB_PATH, SIGNATURES, load_library, make_format,  # noqa: F401
                   shard_offsets, shard_range, source_fingerprint, synth_pcm)

__all__ = ["AlacError", "Comm", "Context", "Format", "LIB_PATH", "SIGNATURES", "load_library", "make_format",
           "shard_offsets", "shard_range", "source_fingerprint", "synth_pcm"]

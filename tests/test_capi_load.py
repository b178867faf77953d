"""CPU: the C-ABI library loads and exports every symbol include/alac_hip.h declares (no compute calls
without a GPU), and the Python binding table mirrors the header."""
import ctypes
import os
import re

import pytest

import alac_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "alac_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(alac_(?:hip|synth)_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    lib = ctypes.CDLL(alac_amd.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/alac_hip.h but not exported"


def test_library_exports_the_reference_stage_surface():
    """SURVEY.md §8b "C stage ABI": every prototype of include/alac/{matrixlib,dplib,aglib,ALACBitUtilities}.h — the
    reference's codec/matrixlib.h:41-60, dplib.h:49-55, aglib.h:70-74, ALACBitUtilities.h:85-97 — is an exported symbol."""
    lib = ctypes.CDLL(alac_amd.LIB_PATH)
    want = {"matrixlib.h": ["mix16", "mix20", "mix24", "mix32", "copy20ToPredictor"],
            "dplib.h": ["init_coefs", "copy_coefs", "pc_block", "unpc_block"],
            "aglib.h": ["set_standard_ag_params", "set_ag_params", "dyn_comp", "dyn_decomp"],
            "ALACBitUtilities.h": ["BitBufferInit", "BitBufferRead", "BitBufferReadSmall", "BitBufferReadOne", "BitBufferPeek",
                                   "BitBufferPeekOne", "BitBufferUnpackBERSize", "BitBufferGetPosition", "BitBufferByteAlign",
                                   "BitBufferAdvance", "BitBufferRewind", "BitBufferWrite", "BitBufferReset"]}
    for header, names in want.items():
        with open(os.path.join(ROOT, "include", "alac", header)) as f:
            text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
        declared = set(re.findall(r"\b([A-Za-z_]\w*)\s*\(", text))
        for n in names:
            assert n in declared, f"{n} not declared in include/alac/{header}"
            assert hasattr(lib, n), f"{n} declared in include/alac/{header} but not exported"


def test_binding_table_matches_header():
    assert sorted(alac_amd.SIGNATURES) == _declared()
    alac_amd.load_library()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        alac_amd.Context(0)


def test_host_only_entry_points():
    lib = alac_amd.load_library()
    fmt = alac_amd.make_format(4096, 16, 2, 44100)
    ck = (ctypes.c_uint8 * 24)()
    assert lib.alac_hip_magic_cookie(ctypes.byref(fmt), 0, 0, ck) == 24
    assert bytes(ck).hex() == "000010000010280a0e0200ff00000000000000000000ac44"
    back = alac_amd.Format()
    assert lib.alac_hip_format_from_cookie(ck, 24, ctypes.byref(back)) == 0
    assert (back.frame_size, back.bit_depth, back.num_channels, back.sample_rate) == (4096, 16, 2, 44100)
    assert lib.alac_hip_format_from_cookie(ck, 8, ctypes.byref(back)) == -50
    bad = alac_amd.make_format(4096, 12, 2)
    assert lib.alac_hip_encode_workspace_bytes(ctypes.byref(bad), 1, 1) == 0
    assert lib.alac_hip_encode_max_output_bytes(ctypes.byref(fmt), 1) >= 16388
    # legacy 'frma' + 'alac' wrappers in front of the config (ALACMagicCookieDescription.txt)
    wrapped = bytes(4) + b"frma" + b"alac" + bytes(4) + b"alac" + bytes(4) + bytes(ck)
    arr = (ctypes.c_uint8 * len(wrapped)).from_buffer_copy(wrapped)
    assert lib.alac_hip_format_from_cookie(arr, len(wrapped), ctypes.byref(back)) == 0 and back.bit_depth == 16


def test_reference_stage_prototypes_are_exported():
    """include/alac/dplib.h and aglib.h: the reference's own host-callable stage symbols (codec/dplib.h:49-55,
    codec/aglib.h:70-74) resolve in the library."""
    import ctypes
    import alac_amd
    lib = ctypes.CDLL(alac_amd.LIB_PATH)
    for name in ("init_coefs", "copy_coefs", "pc_block", "unpc_block", "set_ag_params", "set_standard_ag_params",
                 "dyn_comp", "dyn_decomp"):
        assert getattr(lib, name) is not None


def test_shard_arithmetic_is_host_only_and_tiles_the_units():
    """contiguous ranges per rank that tile [0, S), prefix-sum offsets of the shard bytes"""
    for units, world in [(1000000 // 8 * 8, 8), (10, 3), (5, 8), (0, 2), (125000 * 8, 8), (7, 1)]:
        at = 0
        for r in range(world):
            first, count = alac_amd.shard_range(units, world, r)
            assert first == at and count in (units // world, units // world + 1)
            at += count
        assert at == units
    with pytest.raises(ValueError):
        alac_amd.shard_range(10, 0, 0)
    with pytest.raises(ValueError):
        alac_amd.shard_range(10, 4, 4)
    assert alac_amd.shard_offsets([5, 0, 7]) == [0, 5, 5, 12]
    with pytest.raises(ValueError):
        alac_amd.shard_offsets([2 ** 64 - 1, 2])

"""alac_hip_set_option / alac_hip_get_option / alac_hip_encode_regime (include/alac_hip.h): every documented key round-trips,
an unknown key is refused with kALAC_ParamError, options belong to ONE context, and the regime the library reports follows
them.  (What the options DO is covered by tests/test_gpu_variants.py, which runs the parity files under each.)"""
import pytest

import alac_amd

pytestmark = pytest.mark.gpu

KEYS = {"thru": (-1, 1), "narrow": (-1, 1), "split_coder": (0, 1), "overlap_pos": (0, 1), "fused": (0, 1), "fold": (0, 1),
        "fast_mode": (0, 1), "encoder_lane": (0, 1), "decoder_lane": (0, 1), "dec_fused": (-1, 1), "dec_pair": (0, 1), "dec_direct": (0, 2),
        "stage_taps": (0, 1), "debug_lose_handoff": (0, 1), "debug_waves": (0, 1)}
REMOVED = ["idlefast", "wide81", "pubfence", "subbatch", "persist", "class_fused", "search_fused", "thru_wg4", "lds_pad",
           "count_walk", "init_state", "dec_wide", "dec_local", "dec_pubmask"]  # measured and rejected variants, gone in round 4


def test_every_documented_key_round_trips_and_is_documented():
    import os
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "alac_hip.h")).read()
    ctx = alac_amd.Context(0)
    for k, (lo, hi) in KEYS.items():
        old = ctx.get_option(k)
        for v in list(range(lo, hi + 1)) + [old]:
            ctx.set_option(k, v)
            assert ctx.get_option(k) == v, k
        for v in (lo - 1, hi + 1, 1 << 20, -(1 << 20)):  # values outside the documented range are refused
            with pytest.raises(Exception):
                ctx.set_option(k, v)
            assert ctx.get_option(k) == old, k
        assert f'"{k}"' in header, f"{k} is not listed in include/alac_hip.h"
    for k in REMOVED:
        with pytest.raises(Exception):
            ctx.set_option(k, 0)
        assert f'"{k}"' not in header


def test_unknown_key_is_a_param_error():
    ctx = alac_amd.Context(0)
    with pytest.raises(Exception) as e:
        ctx.set_option("no_such_option", 1)
    assert "-50" in str(e.value) or "Param" in str(e.value)
    with pytest.raises(Exception):
        ctx.get_option("no_such_option")


def test_options_belong_to_one_context_and_steer_the_regime():
    a, b = alac_amd.Context(0), alac_amd.Context(0)
    fmt = alac_amd.make_format(4096, 16, 2, 44100)
    assert a.regime(fmt, 10000) == "latency" and a.regime(fmt, 125000) == "throughput" and a.regime(fmt, 100) == "tiny"
    a.set_option("thru", 1)
    assert a.regime(fmt, 10000) == "throughput" and b.regime(fmt, 10000) == "latency"
    a.set_option("thru", 0)
    assert a.regime(fmt, 125000) != "throughput"
    with b.options(encoder_lane=1):
        assert b.regime(fmt, 10000) == "lane"
    assert b.regime(fmt, 10000) == "latency"
    with b.options(fused=0):  # the launcher's predicate: narrow only counts while the launches are fused
        assert b.regime(fmt, 100) == "stagewise" and b.regime(fmt, 10000) == "stagewise" and b.regime(fmt, 125000) == "throughput"

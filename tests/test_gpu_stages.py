"""GPU parity at stage level: the batched pc_block / unpc_block / dyn_comp / dyn_decomp entry points of
the C-ABI against the golden vectors from the reference's compiled stage objects (incl. the general-path
"deep LPC" tap counts 16 and 30, BASELINE configs[2])."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def stage():
    z = np.load(os.path.join(GOLD, "stage_vectors.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def test_pc_and_unpc_block(gpu_ctx, stage):
    import torch
    z, meta = stage
    seen = set()
    for m in [m for m in meta if m["kind"] == "pc"]:
        i, num, na, cb = m["id"], m["num"], m["numactive"], m["chanbits"]
        seen.add(na)
        x = z[f"pc{i}_x"]
        rows = 3  # same row three times: lanes must not interfere
        dx = torch.from_numpy(np.tile(x, (rows, 1))).cuda()
        co = torch.from_numpy(np.tile(z[f"pc{i}_coefs"], (rows, 1))).cuda()
        pc = gpu_ctx.pc_block(dx, num, co, na, cb)
        gpu_ctx.synchronize()
        for r in range(rows):
            assert np.array_equal(pc[r, :num].cpu().numpy(), z[f"pc{i}_pc"][:num]), m
            if na not in (0, 31):
                assert np.array_equal(co[r, :na].cpu().numpy(), z[f"pc{i}_after"][:na]), m
        co2 = torch.from_numpy(np.tile(z[f"pc{i}_coefs"], (rows, 1))).cuda()
        back = gpu_ctx.pc_block(pc, num, co2, na, cb, decode=True)
        gpu_ctx.synchronize()
        sh = 32 - cb
        want = ((x[:num].astype(np.int64) << sh).astype(np.int32) >> sh) if na else x[:num]
        assert np.array_equal(back[0, :num].cpu().numpy(), want), m
    assert {4, 8, 16, 30, 31, 0} <= seen


def test_dyn_comp_and_decomp(gpu_ctx, stage):
    import torch
    z, meta = stage
    for m in [m for m in meta if m["kind"] == "ag"]:
        j, n, bits = m["id"], m["n"], m["bits"]
        pc = z[f"ag{j}_pc"]
        # the fixture was coded at a start bit offset; the GPU entry point codes from bit 0: compare bits
        dpc = torch.from_numpy(np.tile(pc if n else np.zeros(1, np.int32), (2, 1))).cuda()
        stride = ((n * 8 + 64) + 3) // 4 * 4
        out, nb = gpu_ctx.dyn_comp(dpc, n, bits, stride)
        gpu_ctx.synchronize()
        assert nb.cpu().tolist() == [m["nbits"]] * 2, m
        got = np.unpackbits(out[0].cpu().numpy())[:m["nbits"]]
        want = np.unpackbits(z[f"ag{j}_bytes"])[m["start_bit"]:m["start_bit"] + m["nbits"]]
        assert np.array_equal(got, want), m
        back, nb2, st = gpu_ctx.dyn_decomp(out, n, bits)
        gpu_ctx.synchronize()
        assert st.cpu().tolist() == [0, 0] and nb2.cpu().tolist() == [m["nbits"]] * 2, m
        assert np.array_equal(back[1, :n].cpu().numpy(), pc[:n]), m


@pytest.mark.parametrize("na", [1, 2, 3, 5, 7, 9, 12, 15, 16, 17, 23, 30])
def test_pc_block_any_tap_count_matches_oracle(gpu_ctx, oracle, na):
    """the general loop of pc_block (dp_enc.c:341-387) for every tap count, incl. the tap-parallel kernel (>= 5
    taps: one chain per half wave) — several rows so that both halves of a wave and a second wave are used"""
    import torch
    rng = np.random.default_rng(100 + na)
    for chanbits, num in ((17, 700), (21, 333), (24, 64 + na), (16, na + 2)):
        rows = 5
        amp = 1 << (chanbits - 3)
        t = np.arange(num + 40)
        x = np.stack([(amp * np.sin(t * rng.uniform(0.01, 0.3)) + rng.integers(-amp // 8, amp // 8 + 1, num + 40)).astype(np.int32)
                      for _ in range(rows)])
        co0 = np.zeros((rows, 32), np.int16)
        co0[:, :3] = [1216, -928, -64]
        co0[:, 3:na] = rng.integers(-50, 50, (rows, max(na - 3, 0)))
        co = torch.from_numpy(co0.copy()).cuda()
        pc = gpu_ctx.pc_block(torch.from_numpy(x).cuda(), num, co, na, chanbits)
        gpu_ctx.synchronize()
        for r in range(rows):
            want_pc, want_co = oracle.pc_block(x[r], num, co0[r], na, chanbits)
            assert np.array_equal(pc[r, :num].cpu().numpy(), want_pc[:num]), (na, chanbits, num, r)
            assert np.array_equal(co[r, :na].cpu().numpy(), want_co[:na]), (na, chanbits, num, r)


@pytest.mark.parametrize("chanbits", [24, 25, 32])
def test_wide_full_scale_samples_wrap_like_the_reference(gpu_ctx, oracle, chanbits):
    """full-scale material at chanBits 24 / 25 (24-bit coded without shift-off bytes) and 32 (32-bit mono): differences of
    two samples overflow int32 at 32 bits and the reference's compiled objects wrap (pinned in tests/test_oracle.py); every
    tap count class and denominator shift other than 9 (the stage entry points take the header's fields)"""
    import torch
    rng = np.random.default_rng(chanbits)
    lim = (1 << (chanbits - 1)) - 1
    # (small shifts with 4 / 8 taps: one term of the coefficient walk exceeds 2^31 at chanBits 32 and del0 wraps — the walk must
    # still stop where the reference's stops, foreign-stream soak seed 289)
    for na, ds in ((4, 9), (8, 9), (8, 5), (4, 12), (1, 12), (6, 5), (16, 7), (30, 4), (31, 9), (0, 9), (2, 1), (12, 15), (8, 4),
                   (8, 1), (4, 3), (4, 1), (8, 2)):
        rows, num = 4, 200
        x = rng.integers(-lim - 1, lim + 1, size=(rows, num + 40)).astype(np.int32)
        x[1] //= 3
        co0 = rng.integers(-(1 << ds), (1 << ds) + 1, size=(rows, 32)).clip(-32768, 32767).astype(np.int16)
        co = torch.from_numpy(co0.copy()).cuda()
        pc = gpu_ctx.pc_block(torch.from_numpy(x).cuda(), num, co, na, chanbits, denshift=ds)
        gpu_ctx.synchronize()
        co2 = torch.from_numpy(co0.copy()).cuda()
        back = gpu_ctx.pc_block(pc, num, co2, na, chanbits, denshift=ds, decode=True)
        gpu_ctx.synchronize()
        for r in range(rows):
            want_pc, want_co = oracle.pc_block(x[r], num, co0[r], na, chanbits, ds)
            assert np.array_equal(pc[r, :num].cpu().numpy(), want_pc[:num]), (na, ds, r)
            want_x, want_co2 = oracle.unpc_block(want_pc, num, co0[r], na, chanbits, ds)
            assert np.array_equal(back[r, :num].cpu().numpy(), want_x[:num]), (na, ds, r)
            if na not in (0, 31):
                assert np.array_equal(co[r, :na].cpu().numpy(), want_co[:na]), (na, ds, r)
                assert np.array_equal(co2[r, :na].cpu().numpy(), want_co2[:na]), (na, ds, r)

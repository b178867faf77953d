"""The GPU coder's quotient (alac_golomb.hpp, golf_sym): n / (2^k - 1) as d1 = (n + (n >> k) + 1) >> k,
div = (n + d1 + 1) >> k.  Exact wherever the coder uses it — n < 9 (2^k - 1), the non-escape branch of dyn_code_32bit
(codec/ag_enc.c:151-183) — for k >= 2; for k = 1 it can be one short at n = 7, 8, where the emitted code is the same anyway.
Checked here by enumeration for every k the format allows (kb <= 16)."""
import numpy as np


def _div(n, k):
    k = np.uint64(k)
    d1 = (n + (n >> k) + np.uint64(1)) >> k
    return (n + d1 + np.uint64(1)) >> k


def _code(n, k, div):
    """(numBits, value) of dyn_code_32bit's non-escape branch, the way golf_sym forms them from div"""
    m = np.uint64((1 << k) - 1)
    mod = n - div * m
    ne = np.minimum(mod, np.uint64(1))
    kd = np.uint64(k) + ne
    return div + kd, (((np.uint64(1) << div) - np.uint64(1)) << kd) + mod + ne


def test_quotient_is_exact_below_the_escape_for_k_of_two_and_more():
    for k in range(2, 17):
        m = (1 << k) - 1
        n = np.arange(0, 9 * m, dtype=np.uint64)
        assert np.array_equal(_div(n, k), n // np.uint64(m)), k
        assert int((n + (n >> np.uint64(k)) + np.uint64(1)).max()) < 1 << 32  # 32-bit arithmetic does not wrap


def test_emitted_code_is_the_reference_code_for_every_k():
    for k in range(1, 17):
        m = (1 << k) - 1
        n = np.arange(0, 9 * m, dtype=np.uint64)
        got = _code(n, k, _div(n, k))
        want = _code(n, k, n // np.uint64(m))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), k
    # ... although for k = 1 the quotient itself is one short at n = 7 and 8
    n = np.arange(0, 9, dtype=np.uint64)
    assert (_div(n, 1) != n).sum() == 2

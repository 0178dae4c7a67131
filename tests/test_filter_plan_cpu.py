"""The polar filter's composite-radix plan (csrc/fft_lds.h) replayed on the host: the pass structure the
kernels run -- forward Stockham passes, the merged pass (last forward butterfly + multiplier + first
inverse butterfly in one thread's registers), inverse passes with the radices REVERSED -- with the
library's own radices and umulhi magic numbers (gcm_filter_plan), against numpy.fft, which is what
the reference's low_pass.arakawa_1977 calls (low_pass.py:41-78).  Checks the index arithmetic of
composite_pass / merged_pass, not the butterflies (an exact DFT matrix stands in for them)."""
import ctypes as C

import numpy as np
import pytest


def _plan(n):
    from gcmiipy_amd import _lib
    out = (C.c_uint * 64)()
    assert _lib.lib.gcm_filter_plan(n, out, 64) == 0
    ok, npass, threads = out[0], out[1], out[2]
    passes = [(out[3 + 4 * p], out[4 + 4 * p], out[5 + 4 * p], out[6 + 4 * p]) for p in range(npass)]
    return ok, threads, passes


def _dft(v, inv):
    r = len(v)
    w = np.exp((2j if inv else -2j) * np.pi * np.outer(np.arange(r), np.arange(r)) / r)
    return w @ v


def _pass(x, n, r, ns, magic, inv, threads):
    """composite_pass: thread b < n / r takes x[b + m nb], twiddles by w^(m k nb / Ns), writes j0 + q Ns"""
    nb = n // r
    assert nb <= threads
    y = np.zeros(n, complex)
    for b in range(nb):
        blk = b if ns == 1 else (b * magic) >> 32          # __umulhi(b, magic)
        assert blk == b // ns                              # the magic number divides exactly on the range used
        k = b - blk * ns
        j0 = blk * ns * r + k
        v = np.array([x[b + m * nb] for m in range(r)])
        if ns > 1:
            assert nb % ns == 0
            w = np.exp((2j if inv else -2j) * np.pi * (k * (nb // ns)) / n)
            v = v * w ** np.arange(r)
        v = _dft(v, inv)
        for q in range(r):
            y[j0 + q * ns] = v[q]
    return y


def _merged(x, n, r, s_row, twiddle, threads):
    """merged_pass: last forward pass (Ns = n / r), S[n folded] / n, first inverse pass (Ns = 1)"""
    nb = n // r
    assert nb <= threads
    y = np.zeros(n, complex)
    for b in range(nb):
        v = np.array([x[b + m * nb] for m in range(r)])
        if twiddle:
            v = v * np.exp(-2j * np.pi * b / n) ** np.arange(r)
        v = _dft(v, False)
        for q in range(r):
            f = b + q * nb
            v[q] *= s_row[f if f <= n // 2 else n - f] / n
        v = _dft(v, True)
        for q in range(r):
            y[b * r + q] = v[q]
    return y


def _filter(x, passes, s_row, threads):
    n, npass = len(x), len(passes)
    rad = [a * b for a, b, _, _ in passes]
    ns = 1
    for p in range(npass - 1):
        x = _pass(x, n, rad[p], ns, passes[p][2], False, threads)
        ns *= rad[p]
    assert ns * rad[-1] == n
    x = _merged(x, n, rad[-1], s_row, npass > 1, threads)
    ns = rad[-1]
    for q in range(1, npass):
        p = npass - 1 - q
        x = _pass(x, n, rad[p], ns, passes[q][3], True, threads)
        ns *= rad[p]
    return x


@pytest.mark.parametrize("n", [1440, 2880, 4096, 36, 20, 120, 360, 2250, 400, 1250, 720, 96, 14])
def test_plan_replayed_against_numpy_fft(n):
    ok, threads, passes = _plan(n)
    if n == 14:                                            # a prime factor 7: the generic ping-pong path serves it
        assert not ok
        return
    assert ok and threads % 64 == 0 and threads <= 512
    assert int(np.prod([a * b for a, b, _, _ in passes])) == n
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)   # two real rows packed as re / im
    s_row = rng.random(n // 2 + 1)                             # any real symmetric multiplier
    s_full = np.array([s_row[f if f <= n // 2 else n - f] for f in range(n)])
    want = np.fft.ifft(np.fft.fft(x) * s_full)
    got = _filter(x, passes, s_row, threads)
    assert np.max(np.abs(got - want)) < 1e-12 * max(1.0, np.max(np.abs(want)))
    # the packed rows stay separable: a real symmetric multiplier keeps re and im apart (low_pass.py:61-72)
    re = np.fft.irfft(np.fft.rfft(x.real) * s_row, n)
    im = np.fft.irfft(np.fft.rfft(x.imag) * s_row, n)
    assert np.max(np.abs(got.real - re)) < 1e-12 and np.max(np.abs(got.imag - im)) < 1e-12


def test_plan_shapes_of_the_baseline_grids():
    """1440 = 10.12.12 and 2880 = 15.12.16: three passes, the looping kernels' instantiations rely on it"""
    for n, want, threads in ((1440, [10, 12, 12], 192), (2880, [15, 12, 16], 256)):
        ok, th, passes = _plan(n)
        assert ok and th == threads
        assert [a * b for a, b, _, _ in passes] == want

"""Accuracy of the device algorithm for (p/P0)**kappa, checked on the host: the
table comes from the library (gcm_exner_table), the evaluation below mirrors
gcm_math.h's exner() operation for operation in float64, and the yardstick is
x87 extended precision.  The reference uses numpy's pow (temperature.py:10)."""
import ctypes as C

import numpy as np

KAPPA = 287.0 / 1004.0


def _table():
    from gcmiipy_amd import _lib
    tab = np.empty(256)
    assert _lib.lib.gcm_exner_table(tab.ctypes.data_as(C.POINTER(C.c_double))) == 0
    return tab


def exner_emulated(p, tab):
    bits = p.view(np.int64)
    hi = (bits >> 32).astype(np.int64)
    e = ((hi >> 20) & 0x7ff) - 1023
    idx = (hi >> 14) & 63
    m = ((bits & 0x000fffffffffffff) | 0x3ff0000000000000).view(np.float64)
    E = tab[np.clip(e, -64, 63) + 64]
    rc, ck = tab[128 + 2 * idx], tab[129 + 2 * idx]
    # t = fma(m, rc, -1): exact product minus one; emulate with extended precision
    t = (np.longdouble(m) * np.longdouble(rc) - 1).astype(np.float64)
    b = [KAPPA]
    for n in range(2, 8):
        b.append(b[-1] * (KAPPA - (n - 1)) / n)
    t2 = t * t
    t4 = t2 * t2
    q12 = t * b[1] + b[0]
    q34 = t * b[3] + b[2]
    q56 = t * b[5] + b[4]
    hi3 = t4 * b[6] + (t2 * q56 + q34)
    poly = (t * t2) * hi3 + (t * q12 + 1.0)
    return E * ck * poly


def test_table_entries():
    tab = _table()
    assert tab[64] == float(np.exp(-KAPPA * np.log(np.longdouble(1e5))))     # e = 0
    for i in (0, 17, 63):
        rc = tab[128 + 2 * i]
        assert abs(rc * (1 + (i + 0.5) / 64) - 1) < 2e-16
        want = np.power(1 / np.longdouble(rc), np.longdouble(KAPPA))
        assert abs(tab[129 + 2 * i] / float(want) - 1) < 2e-16


def test_exner_accuracy_against_extended_precision():
    tab = _table()
    rng = np.random.default_rng(0)
    for lo, hi in ((9e4, 1.1e5), (1e2, 1.1e5), (1e-3, 1e12)):
        p = np.exp(rng.uniform(np.log(lo), np.log(hi), 200000))
        ref = np.exp(np.longdouble(KAPPA) * np.log(np.longdouble(p) / np.longdouble(1e5)))
        mine = exner_emulated(p, tab)
        err = float(np.max(np.abs((mine - ref) / ref)))
        npw = float(np.max(np.abs((np.power(p / 1e5, KAPPA) - ref) / ref)))
        assert err < 6e-16, (lo, hi, err, npw)   # measured 2.9e-16 .. 5.3e-16


def test_exner_interval_edges():
    """mantissa-interval boundaries and powers of two: |t| stays <= 2^-7"""
    tab = _table()
    m = 1 + np.arange(0, 65) / 64.0
    p = np.concatenate([m * 2.0 ** k for k in (-3, 0, 16, 17)])
    p = np.concatenate([p, np.nextafter(p, 0), np.nextafter(p, np.inf)])
    ref = np.exp(np.longdouble(KAPPA) * np.log(np.longdouble(p) / np.longdouble(1e5)))
    assert float(np.max(np.abs((exner_emulated(p, tab) - ref) / ref))) < 5e-16

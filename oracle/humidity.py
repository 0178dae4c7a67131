"""Humidity helpers (reference humidity.py:4-60), plain SI arrays.  TEST INFRASTRUCTURE: a NumPy
restatement used by tests/ as the checker; pinned bit for bit by tests/golden/g14_humidity.npz, which
tests/golden/make_golden.py generates by running the reference's own humidity.py."""
import numpy as np

from .constants import Rd, Rv


def manabe_rh(geom):
    """humidity.py:4-7: relative humidity profile of Manabe 1967."""
    return 0.77 * (geom.sig - 0.02) / (1 - 0.02)


def saturation_vapor_pressure(tt):
    """humidity.py:10-14: Buck equation; tt in K -> Pa (the reference's kPa literal is 1000 Pa)."""
    t = tt - 273.15
    return 0.61121 * 1000.0 * np.exp((18.678 - t / 234.5) * (t / (257.14 + t)))


def w_s_at(tp, tt):
    """humidity.py:17-20: saturation mixing ratio."""
    e_s = saturation_vapor_pressure(tt)
    return (Rd / Rv) * e_s / (tp - e_s)


def vmr_from_mmr(mmr, mmg, mma):
    """humidity.py:23-24: volume mixing ratio from mass mixing ratio and the two molar masses."""
    return mma / mmg * mmr


def rh_to_mmr(rh, tp, tt):
    """humidity.py:27-37."""
    e_s = saturation_vapor_pressure(tt)
    e = rh * e_s
    w = e * Rd / (Rv * (tp - e))
    return w / (w + 1)


def mmr_to_rh(mmr, tp, tt):
    """humidity.py:40-60."""
    e_s = saturation_vapor_pressure(tt)
    w = mmr / (1 - mmr)
    e = w * tp / (Rd / Rv + w)
    return e / e_s

"""Drop-in for the reference's humidity.py (:4-60): the moisture helpers the 2.5-D harness uses to
build its initial humidity field (no_limits_2_5d.py:146-168 imports them with `from humidity import *`).
They are evaluated once, on the host, so they are NumPy here as in the reference -- same names, same
argument order, same expression order.  Arguments may be plain SI values / arrays (K, Pa, kg/mol) or
pint-like quantities (anything with `.to_base_units()` and `.m`); results are plain SI magnitudes
(Pa for the vapour pressure, dimensionless otherwise)."""
import numpy as np

from .units import strip

Rd, Rv = 287.0, 461.0                                   # constants.py:16,78


def _si(x):
    return strip(x)[0]


def manabe_rh(geom):
    """humidity.py:4-7: relative humidity profile of Manabe 1967 on the geometry's sigma levels."""
    return 0.77 * (_si(geom.sig) - 0.02) / (1 - 0.02)


def saturation_vapor_pressure(tt):
    """humidity.py:10-14: Buck equation; the reference's `0.61121 kPa` is 611.21 Pa."""
    t = _si(tt) - 273.15
    return 0.61121 * 1000.0 * np.exp((18.678 - t / 234.5) * (t / (257.14 + t)))


def w_s_at(tp, tt):
    """humidity.py:17-20: saturation mixing ratio at true pressure tp and true temperature tt."""
    e_s = saturation_vapor_pressure(tt)
    return (Rd / Rv) * e_s / (_si(tp) - e_s)


def vmr_from_mmr(mmr, mmg, mma):
    """humidity.py:23-24: volume mixing ratio from a mass mixing ratio (molar masses of the gas and of air)."""
    return _si(mma) / _si(mmg) * _si(mmr)


def rh_to_mmr(rh, tp, tt):
    """humidity.py:27-37: relative humidity -> specific humidity q = w / (w + 1)."""
    e = _si(rh) * saturation_vapor_pressure(tt)
    w = e * Rd / (Rv * (_si(tp) - e))
    return w / (w + 1)


def mmr_to_rh(mmr, tp, tt):
    """humidity.py:40-60: the inverse of rh_to_mmr."""
    e_s = saturation_vapor_pressure(tt)
    m = _si(mmr)
    w = m / (1 - m)
    e = w * _si(tp) / (Rd / Rv + w)
    return e / e_s

"""Drop-in for the numeric part of the reference's constants.py: the physical constants the hot path
uses (SI magnitudes; the reference wraps them in pint units, constants.py:16-71) and its two
reductions, `get_total_variation` (:105-108) and `courant_number` (:111-112), computed on the GPU
(`gcm_array_stats`)."""
import numpy as np

from ._lib import lib
from .core import _check, as_f64
from .units import strip, scalar

Rd = 287.0                      # constants.py:16   J / (kg K)
Cp = 1004.0                     # :22
kappa = Rd / Cp                 # :28
P0 = 100000.0                   # :31   Pa
standard_pressure = 101325.0    # :37   Pa
standard_temperature = 273.16   # :38   K
G = 9.8                         # :45   m / s**2
radius = 6.3781e6               # :48   m
mu_air = 18.5 * 1e-6            # :51   Pa s (18.5 uPa s)
Rv = 461.0                      # :78   J / (kg K)


def _stats(q):
    a = as_f64(strip(q)[0], name="q")
    if a.ndim < 1 or a.size == 0:
        raise ValueError("expected a non-empty array")
    out = np.empty(3)
    _check(lib.gcm_array_stats(a.ctypes.data, a.shape[0], a.size // a.shape[0], out.ctypes.data), None)
    return out


def get_total_variation(q):
    """sum |q - roll(q, -1, 0)| (constants.py:105-108)"""
    return float(_stats(q)[0])


def courant_number(p, u, dx, dt):
    """(max u + sqrt(mean(p) G)) dt / dx (constants.py:111-112)"""
    return float((_stats(u)[1] + np.sqrt(_stats(p)[2] * G)) * scalar(dt) / scalar(dx))

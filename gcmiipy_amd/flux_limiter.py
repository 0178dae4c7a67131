"""Drop-in for the reference's flux_limiter.py call surface (flux_limiter.py:10-32): the van Leer
limiter function, the smoothness ratio, and the donor-cell flux / advection step on periodic 1-D
arrays, computed on the GPU behind gcm_flux_limiter.  Results are bit-identical to NumPy's,
including what the `b != 0` (calc_r) and strict `u > 0` (donor_cell_flux) masks select."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib
from .core import as_f64
from .two_d import _ops_check
from .units import strip, scalar, attach


def _run(kind, q, u=None, dx=1.0, dt=0.0):
    qm = as_f64(q, name="q")
    if qm.ndim != 1:
        raise ValueError("flux_limiter works on 1-D arrays (coordinates_1d.py)")
    um = None if u is None else as_f64(u, qm.shape, "u")
    out = np.empty_like(qm)
    _ops_check(lib.gcm_flux_limiter(kind, qm.size, qm.ctypes.data_as(C.c_void_p),
                                    None if um is None else um.ctypes.data_as(C.c_void_p),
                                    float(dx), float(dt), out.ctypes.data_as(C.c_void_p)))
    return out


def van_leer(r):                                                    # flux_limiter.py:10-11
    rm, ru = strip(r)
    a = np.asarray(rm, dtype=np.float64)
    out = _run(_lib.FL_VAN_LEER, a.reshape(-1)).reshape(a.shape)
    return attach(out, ru) if a.ndim else float(out)


def calc_r(q):                                                      # flux_limiter.py:14-20
    qm, qu = strip(q)
    return attach(_run(_lib.FL_CALC_R, qm), qu)


def donor_cell_flux(q, u):                                          # flux_limiter.py:23-27
    (qm, _), (um, _) = strip(q), strip(u)
    return _run(_lib.FL_DONOR_FLUX, qm, um)


def donor_cell_advection(q, u, dx, dt):                             # flux_limiter.py:30-32
    (qm, qu), (um, _) = strip(q), strip(u)
    return attach(_run(_lib.FL_DONOR_ADVECTION, qm, um, scalar(dx), scalar(dt)), qu)

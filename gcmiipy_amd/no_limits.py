"""Drop-in for the reference's 1-D model no_limits.py (BASELINE configs[0]): Matsuno of p, u,
theta, q in momentum form on a periodic line (no_limits.py:50-152), computed on the GPU behind
`gcm_pe1d`.  Same names and argument orders; pint quantities or SI ndarrays in, fresh arrays out."""
import ctypes as C

import numpy as np

from ._lib import lib
from .core import as_f64
from .two_d import _ops_check
from .units import strip, scalar, attach


def _arr4(arrs, shape=None):
    a = [as_f64(strip(x)[0], shape, "putq"[n]) for n, x in enumerate(arrs)]
    if a[0].ndim != 1:
        raise ValueError("no_limits works on 1-D arrays (coordinates_1d.py)")
    return a


def _call(nsteps, half_only, dt, dx, base, stage=None):
    b = _arr4(base)
    n = b[0].size
    b = _arr4(base, (n,))
    s = _arr4(stage, (n,)) if stage is not None else None
    out = [np.empty(n) for _ in range(4)]
    P4 = C.c_void_p * 4
    pb = P4(*[x.ctypes.data for x in b])
    ps = P4(*[x.ctypes.data for x in s]) if s is not None else None
    po = P4(*[x.ctypes.data for x in out])
    _ops_check(lib.gcm_pe1d(n, nsteps, 1 if half_only else 0, scalar(dt), scalar(dx), C.byref(pb),
                            C.byref(ps) if ps is not None else None, C.byref(po)))
    return tuple(attach(o, strip(x)[1]) for o, x in zip(out, base))


def half_timestep(p, u, t, q, sp, su, st, sq, dt, dx):              # no_limits.py:115-147
    return _call(0, True, dt, dx, (p, u, t, q), (sp, su, st, sq))


def matsuno_timestep(p, u, t, q, dt, dx):                          # no_limits.py:150-152
    return _call(1, False, dt, dx, (p, u, t, q))


def run(p, u, t, q, dt, dx, steps):
    """`steps` Matsuno steps with the state resident on the device (the reference's test loops,
    no_limits.py:248-262, without the plotting)"""
    return _call(int(steps), False, dt, dx, (p, u, t, q))


# ---- the operators the step is made of, one by one (no_limits.py:50-112): 1-D arrays, SI magnitudes out
def _op1(kind, arrays, dx=1.0):
    from . import _lib
    a = [as_f64(strip(x)[0], name="operand") for x in arrays]
    if a[0].ndim != 1:
        raise ValueError("no_limits works on 1-D arrays (coordinates_1d.py)")
    a = [as_f64(x, a[0].shape, "operand") for x in a] + [None] * (3 - len(a))
    out = np.empty(a[0].shape)
    ptr = lambda x: None if x is None else x.ctypes.data
    _ops_check(lib.gcm_pe1d_op(kind, a[0].size, scalar(dx), ptr(a[0]), ptr(a[1]), ptr(a[2]), out.ctypes.data))
    return out


def advec_q(u, q, dx): return _op1(0, (u, q), dx)               # :50-62
def calc_pu(u, p): return _op1(1, (u, p))                       # :65-67
def un_pu(pu, p): return _op1(2, (pu, p))                       # :69-70
def advec_p(pu, dx): return _op1(3, (pu,), dx)                  # :73-75
def advec_pu(p, pu, u, dx): return _op1(4, (p, pu, u), dx)      # :78-92
def advec_t(pu, t, dx): return _op1(5, (pu, t), dx)             # :95-97
def pgf(p, t, dx): return _op1(6, (p, t), dx)                   # :102-112

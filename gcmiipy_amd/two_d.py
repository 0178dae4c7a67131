"""Drop-in for the reference's two_d.py call surface: tracer / mass schemes on velocity
stacks V[axis] (V[0] acts along array axis 0 with spatial_change[0], two_d.py:16-22), the
C-grid pressure-gradient helpers and the state-dict driver -- computed by the HIP kernels
behind gcm_advect2d / gcm_pgf2d."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib
from .core import GcmError, as_f64
from .units import strip, scalar, attach


def _ops_check(rc):
    if rc == _lib.OK:
        return
    msg = lib.gcm_ops_last_error().decode()
    if rc == _lib.ERR_ARG:
        raise ValueError(msg)
    raise GcmError("gcmcore error %d: %s" % (rc, msg))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _advect(scheme, axes, finite, dt, spatial_change, V, q, nsteps=1):
    (qm, qu), (vm, _) = strip(q), strip(V)
    qm = as_f64(qm, name="q")
    if qm.ndim == 1:                                     # 1-D stacks (test_oneD.py): one row per cell
        vm = as_f64(vm, (1,) + qm.shape, "V")
        out = _advect(scheme, 1, finite, dt, (spatial_change[0], spatial_change[0]),
                      np.stack([vm[0][:, None], np.zeros((qm.size, 1))]), qm[:, None], nsteps)
        return attach(np.asarray(out)[:, 0], qu)
    vm = as_f64(vm, (2,) + qm.shape, "V")
    out = np.empty_like(qm)
    dx0, dx1 = scalar(spatial_change[0]), scalar(spatial_change[1])
    _ops_check(lib.gcm_advect2d(scheme, axes, 1 if finite else 0, qm.shape[1], qm.shape[0], nsteps,
                                scalar(dt), dx0, dx1, _p(vm), _p(qm), _p(out)))
    return attach(out, qu)


def upwind_axis(dt, spatial_change, V, q, axis=0):                      # two_d.py:11-32
    return _advect(_lib.ADV_UPWIND, 1 << axis, False, dt, spatial_change, V, q)


def upwind_axis_finite(dt, spatial_change, V, q, axis=0):               # two_d.py:35-55
    return _advect(_lib.ADV_UPWIND, 1 << axis, True, dt, spatial_change, V, q)


def corner_transport_2d(dt, spatial_change, V, q):                      # two_d.py:59-71
    return _advect(_lib.ADV_UPWIND, 3, False, dt, spatial_change, V, q)


def fv_advect_axis_upwind(dt, spatial_change, V, p, axis=0):            # two_d.py:103-116
    return _advect(_lib.ADV_FV_UPWIND, 1 << axis, False, dt, spatial_change, V, p)


def fv_advect_axis_upwind_finite(dt, spatial_change, V, p, axis=0):     # two_d.py:119-132
    return _advect(_lib.ADV_FV_UPWIND, 1 << axis, True, dt, spatial_change, V, p)


def fv_advect_axis_plain(dt, spatial_change, V, p, axis=0):             # two_d.py:135-149
    return _advect(_lib.ADV_FV_PLAIN, 1 << axis, False, dt, spatial_change, V, p)


def fv_advect_axis_plain_finite(dt, spatial_change, V, p, axis=0):      # two_d.py:152-166
    return _advect(_lib.ADV_FV_PLAIN, 1 << axis, True, dt, spatial_change, V, p)


def finite_volume_advection(dt, spatial_change, V, p, steps=1):         # two_d.py:198-207
    return _advect(_lib.ADV_FV_UPWIND, 3, False, dt, spatial_change, V, p, steps)


def limited_advection(dt, spatial_change, V, q, steps=1):
    """dimension-split finite-volume step with the van_leer(calc_r)-limited centred flux
    (flux_limiter.py:10-27 + two_d.py:103-149; composition by this build)."""
    return _advect(_lib.ADV_VANLEER, 3, False, dt, spatial_change, V, q, steps)


def advect_with_momentum(dt, spatial_change, V, p):                     # two_d.py:277-292
    return _advect(_lib.ADV_MOMENTUM, 3, False, dt, spatial_change, V, p)


def _pgf(kind, dt, spatial_change, p, t=None):
    (pm, _), (tm, _) = strip(p), strip(t)
    pm = as_f64(pm, name="p")
    if pm.ndim != 2:
        raise ValueError("p must be 2-D")
    if len(spatial_change) != 2:
        raise ValueError("spatial_change must have one entry per axis")
    tm = None if tm is None else as_f64(tm, pm.shape, "t")
    out = np.empty((2,) + pm.shape)
    _ops_check(lib.gcm_pgf2d(kind, pm.shape[1], pm.shape[0], scalar(dt), scalar(spatial_change[0]),
                             scalar(spatial_change[1]), _p(pm), _p(tm), _p(out)))
    return out


def pgf_c_grid_axis(p, spatial_change, axis=0):                         # two_d.py:210-220
    return _pgf(0, 0.0, spatial_change, p)[axis]


def pgf_c_grid(dt, spatial_change, p, t):                               # two_d.py:223-245
    return _pgf(1, dt, spatial_change, p, t)


def pgf_templess(dt, spatial_change, p):                                # two_d.py:248-261
    return _pgf(2, dt, spatial_change, p)


def pressure_at_edge(p):                                                # two_d.py:264-268
    return _pgf(3, 0.0, (1.0, 1.0), p)


def _line_or_plane(p):
    """the *_one_d helpers take 1-D lines or 2-D arrays (axis 0 is the one they roll)"""
    pm = as_f64(strip(p)[0], name="p")
    if pm.ndim == 1:
        return pm.reshape(-1, 1), True
    if pm.ndim != 2:
        raise ValueError("p must be 1-D or 2-D")
    return pm, False


def pressure_at_edge_one_d(p):                                          # two_d.py:271-274
    pm, line = _line_or_plane(p)
    out = _pgf(3, 0.0, (1.0, 1.0), pm)[0]
    return out[:, 0] if line else out


def pgf_one_d(dt, dx, p, axis=0):                                       # two_d.py:295-303
    pm, line = _line_or_plane(p)
    if axis not in ((0,) if line else (0, 1)):
        raise ValueError("axis out of range")
    out = _pgf(6, dt, (dx, dx), pm)[axis]
    return out[:, 0] if line else out


def gradient(p, spatial_change, axis):                                  # two_d.py:74-77
    return _pgf(4, 0.0, spatial_change, p)[axis]


def pressure_gradient(dt, spatial_change, p, t):                        # two_d.py:80-100
    return _pgf(5, dt, spatial_change, p, t)


def run_2d_with_ft(initial_conditions, ft, steps=400, display_key="q", variation_key="q",
                   history=None):
    """two_d.py:306-346 without the Matplotlib window: `state = ft(**state)` for `steps`
    steps with the total-variation watch (constants.py:105-108).  Returns True like the
    reference (whose early `return False` is commented out, :338); pass a list as `history`
    to receive the TV series; the final state is in `run_2d_with_ft.last_state`."""
    def tv(q):
        m = np.asarray(strip(q)[0], dtype=np.float64)
        return float(np.sum(np.abs(m - np.roll(m, -1, 0))))

    current = initial_conditions
    initial_variation = tv(current[variation_key])
    if history is not None:
        history.append(initial_variation)
    for _ in range(steps):
        current = ft(**current)
        v = tv(current[variation_key])
        if history is not None:
            history.append(v)
    run_2d_with_ft.last_state = current
    return True

"""Drop-in for the reference's matsumo_temp.py call surface (2-D shallow water +
potential temperature + viscosity), with the optional passive tracer of config 3."""
from . import _lib
from .core import as_f64
from .matsuno_c_grid import _core
from .units import strip, scalar, attach

TRACERS = {None: _lib.TRACER_NONE, "none": _lib.TRACER_NONE, "upwind": _lib.TRACER_UPWIND,
           "van_leer": _lib.TRACER_VANLEER}


def matsumo_scheme(u, v, p, t, dx, dt):
    """matsumo_temp.py:66-99 -- takes and returns (u, v, p, t)."""
    (um, uu), (vm, vu), (pm, pu), (tm, tu) = strip(u), strip(v), strip(p), strip(t)
    um = as_f64(um, name="u")
    if um.ndim != 2:
        raise ValueError("u must be 2-D [j, i]")
    vm, pm, tm = as_f64(vm, um.shape, "v"), as_f64(pm, um.shape, "p"), as_f64(tm, um.shape, "t")
    c = _core(_lib.SW2D_TEMP, um.shape, scalar(dx))
    c.set_state(p=pm, u=um, v=vm, t=tm)
    c.step(1, scalar(dt))
    pn, un, vn, tn, _ = c.get_state((_lib.P, _lib.U, _lib.V, _lib.T))
    return attach(un, uu), attach(vn, vu), attach(pn, pu), attach(tn, tu)


def matsumo_scheme_with_tracer(u, v, p, t, q, dx, dt, scheme="van_leer"):
    """Config-3 step: matsumo_scheme plus a passive tracer q advected by the time-n
    winds with the dimension-split finite-volume step of two_d.py:198-207
    (`scheme="upwind"`) or its van-Leer-limited composition (`"van_leer"`)."""
    (um, uu), (vm, vu), (pm, pu), (tm, tu), (qm, qu) = (strip(x) for x in (u, v, p, t, q))
    um = as_f64(um, name="u")
    vm, pm, tm, qm = (as_f64(x, um.shape, n) for x, n in ((vm, "v"), (pm, "p"), (tm, "t"), (qm, "q")))
    c = _core(_lib.SW2D_TEMP, um.shape, scalar(dx), tracer=TRACERS[scheme])
    c.set_state(p=pm, u=um, v=vm, t=tm, q=qm)
    c.step(1, scalar(dt))
    pn, un, vn, tn, qn = c.get_state()
    return attach(un, uu), attach(vn, vu), attach(pn, pu), attach(tn, tu), attach(qn, qu)


def run_with_callbacks(i, u, v, p, t, dx, dt, callbacks=None):
    """matsumo_temp.py:110-118."""
    u, v, p, t = matsumo_scheme(u, v, p, t, dx, dt)
    for callback in callbacks or []:
        callback(u, v, p, t, i)
    return u, v, p, t


# the operators the step is made of, one by one (matsumo_temp.py:13-47): SI magnitudes out
from .operators import (density_from, potential_temperature, scaling, unscaling, advect_t,  # noqa: E402,F401
                        geopotential_from)

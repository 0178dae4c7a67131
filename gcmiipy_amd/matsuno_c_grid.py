"""Drop-in for the reference's matsuno_c_grid.py call surface (2-D shallow water,
Matsuno on the doubly periodic C-grid), computed by the HIP kernels."""
import numpy as np

from . import _lib
from .core import Core, as_f64
from .units import strip, scalar, attach

_cache = {}


def _core(model, shape, dx, **kw):
    key = (model, shape, dx, tuple(sorted(kw.items())))
    c = _cache.get(key)
    if c is None:
        if len(_cache) > 8:
            _cache.popitem()[1].close()
        c = _cache[key] = Core(model, shape[1], shape[0], dx=dx, **kw)
    return c


def matsumo_scheme(u, v, p, dx, dt):
    """matsuno_c_grid.py:125-142 -- one Matsuno step; takes and returns (u, v, p).
    Inputs are not modified; fresh arrays come back (re-wrapped in the inputs'
    base units when Quantities came in)."""
    (um, uu), (vm, vu), (pm, pu) = strip(u), strip(v), strip(p)
    um = as_f64(um, name="u")
    if um.ndim != 2:
        raise ValueError("u must be 2-D [j, i]")
    vm, pm = as_f64(vm, um.shape, "v"), as_f64(pm, um.shape, "p")
    c = _core(_lib.SW2D, um.shape, scalar(dx))
    c.set_state(p=pm, u=um, v=vm)
    c.step(1, scalar(dt))
    pn, un, vn, _, _ = c.get_state((_lib.P, _lib.U, _lib.V))
    return attach(un, uu), attach(vn, vu), attach(pn, pu)


def courant_number(p, u, dx, dt):
    """matsuno_c_grid.py:121-122 / constants.py:111-112, by device reductions."""
    (um, _), (pm, _) = strip(u), strip(p)
    um = as_f64(um)
    c = _core(_lib.SW2D, um.shape, scalar(dx))
    c.set_state(p=as_f64(pm, um.shape), u=um, v=np.zeros_like(um))
    return (c.diag(_lib.DIAG_MAX_U) + np.sqrt(c.diag(_lib.DIAG_MEAN_P) * 9.8)) * scalar(dt) / scalar(dx)


def run(u, v, p, dx, dt, steps, callback=None, every=1):
    """Device-resident driver loop (the reference's main(), matsuno_c_grid.py:168-187,
    without plotting): `steps` Matsuno steps, NaN watch by device reduction,
    optional callback(i, u, v, p) every `every` steps."""
    (um, uu), (vm, vu), (pm, pu) = strip(u), strip(v), strip(p)
    um = as_f64(um)
    c = Core(_lib.SW2D, um.shape[1], um.shape[0], dx=scalar(dx))
    try:
        c.set_state(p=as_f64(pm, um.shape), u=um, v=as_f64(vm, um.shape))
        done = 0
        while done < steps:
            n = min(every, steps - done) if callback else steps - done
            c.step(n, scalar(dt))
            done += n
            if callback:
                pn, un, vn, _, _ = c.get_state((_lib.P, _lib.U, _lib.V))
                callback(done, attach(un, uu), attach(vn, vu), attach(pn, pu))
            if c.diag(_lib.DIAG_ANY_NAN):
                break
        pn, un, vn, _, _ = c.get_state((_lib.P, _lib.U, _lib.V))
    finally:
        c.close()
    return attach(un, uu), attach(vn, vu), attach(pn, pu)


# the operators the step is made of, one by one (matsuno_c_grid.py:15-118): SI magnitudes out
from .operators import (advection_of_velocity_u, advection_of_velocity_v, geopotential_gradient_u,  # noqa: E402,F401
                        geopotential_gradient_v, advection_of_geopotential)

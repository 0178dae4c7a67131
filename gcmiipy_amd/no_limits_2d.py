"""Drop-in for the reference's no_limits_2d.py call surface (2-D single-layer primitive
equations in momentum form, scalar dx), computed by the HIP kernels (GCM_PE2D)."""
from . import _lib
from .core import as_f64
from .matsuno_c_grid import _core
from .units import strip, scalar, attach


def _prep(p, u, v, t, q):
    vals, units = zip(*(strip(x) for x in (p, u, v, t, q)))
    p0 = as_f64(vals[0], name="p")
    if p0.ndim != 2:
        raise ValueError("p must be 2-D [j, i]")
    return [p0] + [as_f64(a, p0.shape, n) for a, n in zip(vals[1:], "uvtq")], units


def half_timestep(p, u, v, t, q, sp, su, sv, st, sq, dt, dx):
    """no_limits_2d.py:104-126: one Euler stage from base with tendencies on the stage state."""
    base, units = _prep(p, u, v, t, q)
    stage, _ = _prep(sp, su, sv, st, sq)
    c = _core(_lib.PE2D, base[0].shape, scalar(dx))
    c.set_state(*base)
    c.set_star(*stage)
    c.half_step(1, scalar(dt))
    return tuple(attach(a, un) for a, un in zip(c.get_state(), units))


def matsuno_timestep(p, u, v, t, q, dt, dx):
    """no_limits_2d.py:129-131: takes and returns (p, u, v, t, q); q passes through."""
    base, units = _prep(p, u, v, t, q)
    c = _core(_lib.PE2D, base[0].shape, scalar(dx))
    c.set_state(*base)
    c.step(1, scalar(dt))
    return tuple(attach(a, un) for a, un in zip(c.get_state(), units))

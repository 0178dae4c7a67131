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


# ---- the operators the step is made of, one by one (no_limits_2d.py:21-101): 2-D [j, i] arrays,
# scalar dx, SI magnitudes out
def _plane(kind, pair_first, a, b):
    """calc_* / un_* through the 3-D operator of the same name with one level"""
    from . import dynamics
    import numpy as np
    a2, b2 = as_f64(strip(a)[0], name="field"), as_f64(strip(b)[0], name="field")
    if a2.ndim != 2 or a2.shape != b2.shape:
        raise ValueError("expected two 2-D arrays of one shape")
    f = {"calc_pu": dynamics.calc_pu, "calc_pv": dynamics.calc_pv, "un_pu": dynamics.un_pu, "un_pv": dynamics.un_pv}[kind]
    return f(a2, b2[None])[0] if pair_first else f(a2[None], b2)[0]


def calc_pu(p, u): return _plane("calc_pu", True, p, u)                    # :21-23
def calc_pv(p, v): return _plane("calc_pv", True, p, v)                    # :26-28
def un_pu(pu, p): return _plane("un_pu", False, pu, p)                     # :31-33
def un_pv(pv, p): return _plane("un_pv", False, pv, p)                     # :36-38


def advec_p(pu, pv, dx):                                                   # :41-44
    from .operators import stencil
    return stencil(_lib.OP_PE2D_ADVEC_P, (pu, pv), dx)


def advec_m(p, u, v, dx):                                                  # :47-76
    from .operators import stencil
    return stencil(_lib.OP_PE2D_DUT, (p, u, v), dx), stencil(_lib.OP_PE2D_DVT, (p, u, v), dx)


def pgf(p, t, dx):                                                         # :79-92
    from .operators import stencil
    return stencil(_lib.OP_PE2D_PGF_U, (p, t), dx), stencil(_lib.OP_PE2D_PGF_V, (p, t), dx)


def advec_t(pu, pv, t, dx):                                                # :95-101
    from . import dynamics
    import numpy as np
    a = [as_f64(strip(x)[0], name="field") for x in (pu, pv, t)]
    if a[0].ndim != 2:
        raise ValueError("expected 2-D arrays")
    g = dynamics._Shape(1, *a[0].shape)
    g.dx_j = g.dx_h = np.full(a[0].shape[0], scalar(dx))
    g.dy = scalar(dx)
    return dynamics.advec_t(*[x[None] for x in a], g)[0]

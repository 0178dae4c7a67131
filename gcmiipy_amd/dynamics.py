"""Drop-in for the reference's dynamics.py call surface (2.5-D sigma-level primitive
equations): `matsuno_timestep` and `half_timestep`, computed by the HIP kernels."""
import numpy as np

from . import _lib
from .core import Core, as_f64
from .units import strip, scalar, attach

_cache = {}


def _geom_key(geom):
    return (geom.height, geom.width, geom.layers, float(geom.dy), float(geom.ptop),
            np.asarray(geom.dx_j).tobytes(), np.asarray(geom.sig).tobytes(),
            np.asarray(geom.heightmap).tobytes())


def core_for(geom, filter=True, coriolis=False):
    key = (_geom_key(geom), filter, coriolis)
    c = _cache.get(key)
    if c is None:
        if len(_cache) > 4:
            _cache.popitem()[1].close()
        c = _cache[key] = Core(_lib.PE25D, geom.width, geom.height, geom.layers, geom=geom,
                               filter=filter, coriolis=coriolis)
    return c


def _prep(p, u, v, t, q, geom):
    shp3, shp2 = (geom.layers, geom.height, geom.width), (geom.height, geom.width)
    vals, units = zip(*(strip(x) for x in (p, u, v, t, q)))
    arrs = [as_f64(vals[0], shp2, "p")] + [as_f64(a, shp3, n) for a, n in zip(vals[1:], "uvtq")]
    return arrs, units


def _wrap_out(arrs, units):
    return tuple(attach(a, un) for a, un in zip(arrs, units))


def half_timestep(p, u, v, t, q, sp, su, sv, st, sq, dt, geom):
    """dynamics.py:183-227: one Euler stage from base (p,u,v,t,q) with tendencies evaluated
    on the stage state; returns fresh (p_n, u_n, v_n, t_n, q_n)."""
    base, units = _prep(p, u, v, t, q, geom)
    stage, _ = _prep(sp, su, sv, st, sq, geom)
    c = core_for(geom)
    c.set_state(*base)
    c.set_star(*stage)
    c.half_step(1, scalar(dt))          # corrector form: base + stage -> new current state
    return _wrap_out(c.get_state(), units)


def matsuno_timestep(p, u, v, t, q, dt, geom, boundary_conditions=None, coriolis=False):
    """dynamics.py:230-237.  `coriolis=True` switches on the Coriolis terms the reference keeps
    behind `if False` (dynamics.py:82-92).  With a Python `boundary_conditions(sp,su,sv,st,sq,dt,geom)` hook
    the predicted state makes a host round trip between the stages (documented slow path);
    with None both stages stay on the device."""
    base, units = _prep(p, u, v, t, q, geom)
    c = core_for(geom, coriolis=coriolis)
    c.set_state(*base)
    dts = scalar(dt)
    if boundary_conditions is None:
        c.step(1, dts)
        return _wrap_out(c.get_state(), units)
    c.half_step(0, dts)
    star = _wrap_out(c.get_star((0, 1, 2, 3, 4)), units)
    star = boundary_conditions(*star, dt, geom)
    c.set_star(*_prep(*star, geom)[0])
    c.half_step(1, dts)
    out = _wrap_out(c.get_state(), units)
    return boundary_conditions(*out, dt, geom)


def run(p, u, v, t, q, dt, geom, steps, callback=None, every=1):
    """Device-resident loop: `steps` Matsuno steps with the state in HBM throughout;
    optional callback(p,u,v,t,q) every `every` steps (no_limits_2_5d.py:230-234)."""
    base, units = _prep(p, u, v, t, q, geom)
    c = Core(_lib.PE25D, geom.width, geom.height, geom.layers, geom=geom)
    try:
        c.set_state(*base)
        done = 0
        while done < steps:
            n = min(every, steps - done) if callback else steps - done
            c.step(n, scalar(dt))
            done += n
            if callback:
                callback(*_wrap_out(c.get_state(), units))
        return _wrap_out(c.get_state(), units)
    finally:
        c.close()

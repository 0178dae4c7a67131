"""Drop-in for the caller side of the 2.5-D path (reference no_limits_2_5d.py): initial
conditions, `full_timestep` with its STATS record, `calc_energy`, `run_model`.  The time step
and the diagnostics run on the GPU; initial conditions are host NumPy (they are evaluated once)."""
from collections import defaultdict, namedtuple

import numpy as np

from . import _lib, geometry
from .core import Core
from .dynamics import _prep, _wrap_out, core_for
from .units import scalar
from .grey_solar import solar_timestep  # noqa: F401  (no_limits_2_5d.py:66-75 lives beside full_timestep in the reference)
from .humidity import manabe_rh, saturation_vapor_pressure, rh_to_mmr, w_s_at, vmr_from_mmr, mmr_to_rh  # noqa: F401  (no_limits_2_5d.py:28: `from humidity import *`)

Rd, Rv, P0, KAPPA = 287.0, 461.0, 100000.0, 287.0 / 1004.0     # constants.py:16,78,31,28

GroundVars = namedtuple("GroundVars", ("gt", "gw", "snow", "ice"))   # no_limits_2_5d.py:143
STATS = defaultdict(list)                                            # no_limits_2_5d.py:63


def gen_initial_conditions(geom):
    """no_limits_2_5d.py:146-168 -> (p, u, v, t, q, GroundVars)."""
    full = (geom.layers, geom.height, geom.width)
    surface = (geom.height, geom.width)
    p = np.full(surface, 1) * 100000 * 1.0 - geom.ptop
    u = np.full(full, 1) * 1.0
    v = np.full(full, 1) * .0
    tt = np.full(full, 1) * 360 * 1.0
    tp = p * geom.sig + geom.ptop
    t = tt * ((P0 / tp) ** KAPPA)                                    # to_potential_temp
    q = np.maximum(np.full(full, 1) * 0.000003, rh_to_mmr(manabe_rh(geom), tp, tt))
    g = GroundVars(np.full(surface, 1) * 360 * 1.0, np.zeros(surface), np.zeros(surface), np.zeros(surface))
    return p, u, v, t, q, g


def calc_energy(p, u, v, t, q, g, geom):
    """no_limits_2_5d.py:35-60 -> (ke, ate, geo, total), by a device reduction."""
    base, _ = _prep(p, u, v, t, q, geom)
    c = core_for(geom)
    c.set_state(*base)
    return c.energy(geom.area)


def _record(c, geom, stats):
    """no_limits_2_5d.py:85-91: max/min of u and v and calc_energy, one fused device reduction"""
    r = c.stats(geom.area)
    for k in ("u_max", "u_min", "v_max", "v_min", "ke"):
        stats[k].append(r[k])


def full_timestep(p, u, v, t, q, g, dt, utc, geom, stats=STATS):
    """no_limits_2_5d.py:79-94: one Matsuno step + the STATS record; `g` passes through (the
    physics below the reference's early return never runs)."""
    base, units = _prep(p, u, v, t, q, geom)
    c = core_for(geom)
    c.set_state(*base)
    c.step(1, scalar(dt))
    _record(c, geom, stats)
    return (*_wrap_out(c.get_state(), units), g)


def run_model(height, width, layers, dt, timesteps, callback, stats=STATS, bump=None, physics=False):
    """no_limits_2_5d.py:220-236 (and test_geography.py:6-23 with `bump=(j, i, metres)`): the
    state stays in HBM for all `timesteps`; STATS come from device reductions every step.
    physics=True: every step is followed by solar_timestep(t, p, g, dt, utc, geom) with utc = 0, dt, 2 dt, ...
    -- the lines the reference keeps below full_timestep's early return (:93-96) and run_model's clock (:222, :231);
    on the device both phases are one gcm_step (gcm_set_physics), and the returned g carries the new ground temperature."""
    geom = geometry.gen_geometry(height, width, layers, sig_func=geometry.manabe_sig)
    if bump is not None:
        geom.heightmap[bump[0], bump[1]] = bump[2]
    p, u, v, t, q, g = gen_initial_conditions(geom)
    v[0, 0, 0] = 0.1
    u *= 0
    c = Core(_lib.PE25D, width, height, layers, geom=geom)
    try:
        c.set_state(p, u, v, t, q)
        if physics:
            c.set_ground(g.gt)
            c.set_physics(geom, 0.0)
        for _ in range(timesteps):
            c.step(1, scalar(dt))
            _record(c, geom, stats)
            if callback:
                callback(*c.get_state())
        p, u, v, t, q = c.get_state()
        if physics:
            g = GroundVars(c.get_ground(), g.gw, g.snow, g.ice)
    finally:
        c.close()
    return p, u, v, t, q, g, geom

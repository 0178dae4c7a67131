"""Caller / harness of the 2.5-D path (reference no_limits_2_5d.py:35-94,
146-168, 220-236; humidity.py:4-37; two_d.py:306-346)."""
from collections import defaultdict, namedtuple

import numpy as np

from . import geometry
from .constants import Rd, Rv, Cp, G
from .dynamics import matsuno_timestep
from .grid import imh, jmh, get_total_variation
from .temperature import to_true_temp, to_potential_temp

GroundVars = namedtuple("GroundVars", ("gt", "gw", "snow", "ice"))   # no_limits_2_5d.py:143
STATS = defaultdict(list)                                            # :63


def manabe_rh(geom):
    """humidity.py:4-7."""
    return 0.77 * (geom.sig - 0.02) / (1 - 0.02)


def saturation_vapor_pressure(tt):
    """humidity.py:10-14 (Buck; kPa literal -> Pa)."""
    t = tt - 273.15
    return 0.61121 * 1000.0 * np.exp((18.678 - t / 234.5) * (t / (257.14 + t)))


def rh_to_mmr(rh, tp, tt):
    """humidity.py:27-37."""
    e_s = saturation_vapor_pressure(tt)
    e = rh * e_s
    w = e * Rd / (Rv * (tp - e))
    return w / (w + 1)


def gen_initial_conditions(geom):
    """no_limits_2_5d.py:146-168."""
    full = (geom.layers, geom.height, geom.width)
    surface = (geom.height, geom.width)
    p = np.full(surface, 1) * 100000 * 1.0 - geom.ptop
    u = np.full(full, 1) * 1.0 * 1.0 / 1.0
    v = np.full(full, 1) * .0 * 1.0 / 1.0
    tt = np.full(full, 1) * 360 * 1.0
    tp = p * geom.sig + geom.ptop
    t = to_potential_temp(tt, tp)
    q = np.full(full, 1) * 0.000003 * 1.0 * 1.0 ** -1
    q = np.maximum(q, rh_to_mmr(manabe_rh(geom), tp, tt))
    gt = np.full(surface, 1) * 360 * 1.0
    g = GroundVars(gt, np.zeros(surface), np.zeros(surface), np.zeros(surface))
    return p, u, v, t, q, g


def calc_energy(p, u, v, t, q, g, geom):
    """no_limits_2_5d.py:35-60 -> (ke, ate, geo, total)."""
    u_at_center = imh(u)
    v_at_center = jmh(v)
    mag = np.sqrt(u_at_center ** 2 + v_at_center ** 2)
    tp = p * geom.sig + geom.ptop
    tt = to_true_temp(t, tp)
    rho = tp / (Rd * tt)
    dp = p * geom.dsig
    geopotential_depth = dp / (rho * G)
    airmass = rho * geopotential_depth * geom.area
    total_depth = np.cumsum(geopotential_depth, 0)
    geopotential = total_depth * airmass * G
    geo = np.sum(geopotential)
    ke = mag ** 2 * .5 * airmass
    ke = np.sum(ke)
    ate = tt * Cp * airmass
    ate = np.sum(ate)
    return ke, ate, geo, ke + ate + geo


def full_timestep(p, u, v, t, q, g, dt, utc, geom, stats=STATS):
    """no_limits_2_5d.py:79-94 (physics after the early return never runs)."""
    p, u, v, t, q = matsuno_timestep(p, u, v, t, q, dt, geom)
    stats["u_max"].append(np.max(u))
    stats["u_min"].append(np.min(u))
    stats["v_max"].append(np.max(v))
    stats["v_min"].append(np.min(v))
    stats["ke"].append(calc_energy(p, u, v, t, q, g, geom))
    return p, u, v, t, q, g


def run_model(height, width, layers, dt, timesteps, callback, stats=STATS,
              bump=None):
    """no_limits_2_5d.py:220-236; `bump=(j, i, metres)` adds the
    test_geography.py:13 topography bump."""
    geom = geometry.gen_geometry(height, width, layers, sig_func=geometry.manabe_sig)
    if bump is not None:
        geom.heightmap[bump[0], bump[1]] = bump[2]
    p, u, v, t, q, g = gen_initial_conditions(geom)
    utc = 0.0
    v[0, 0, 0] = 0.1
    u *= 0
    for _ in range(timesteps):
        p, u, v, t, q, g = full_timestep(p, u, v, t, q, g, dt, utc, geom, stats)
        utc += dt
        if callback:
            callback(p, u, v, t, q)
    return p, u, v, t, q, g, geom


def run_2d_with_ft(initial_conditions, ft, steps=400, display_key="q",
                   variation_key="q"):
    """two_d.py:306-346 without the plotting; returns (always-True, final
    state, TV history) -- the reference returns only the always-True flag."""
    current = initial_conditions
    tv = [get_total_variation(current[variation_key])]
    for _ in range(steps):
        current = ft(**current)
        tv.append(get_total_variation(current[variation_key]))
    return True, current, tv

"""Drop-in for the reference's grey-atmosphere column physics (grey_solar.py) and
no_limits_2_5d.solar_timestep, computed by the HIP column kernel."""
import math
from collections import namedtuple

import numpy as np

from .dynamics import core_for
from .core import as_f64
from .units import strip, scalar, attach

GroundVars = namedtuple("GroundVars", ("gt", "gw", "snow", "ice"))


def _gt(g):
    return strip(g.gt if hasattr(g, "gt") else g)[0]


def solar_zenith_angle(latitude, hour_angle, declination):
    """grey_solar.py:39-46: cos(zenith angle) from latitude, hour angle and declination (radians).  Host NumPy,
    for callers that want the field itself; the column kernel evaluates the same expression per column."""
    return np.sin(latitude) * np.sin(declination) + np.cos(latitude) * np.cos(declination) * np.cos(hour_angle)


def zenith_angle(longs, lats, time, geom):
    """grey_solar.py:49-65: max(cos(zenith), 0) on the (height, width) grid at `time` seconds UTC."""
    hour_angle = scalar(time) / (-24 * 3600.0) * 360 * (math.pi / 180)
    t_longs = np.tile(strip(longs)[0], (geom.height, 1))
    point_angle = t_longs + hour_angle
    return np.maximum(solar_zenith_angle(strip(lats)[0], point_angle, 0 * (math.pi / 180)), 0)


def basic_grey_radiation(p, tp, tt, g, t_lw, t_sw, albedo, utc, geom):
    """grey_solar.py:358-563 -> (dTdt, dt_ground).  `tp` is implied by p and geom and is not
    read; `tt` is the true temperature (converted back to theta for the resident state)."""
    shp3, shp2 = (geom.layers, geom.height, geom.width), (geom.height, geom.width)
    pm = as_f64(strip(p)[0], shp2, "p")
    ttm = as_f64(strip(tt)[0], shp3, "tt")
    theta = ttm * ((100000.0 / (pm * geom.sig + geom.ptop)) ** (287.0 / 1004.0))   # temperature.py:15-19
    c = core_for(geom)
    z = as_f64(theta * 0.0, shp3)
    c.set_state(pm, z, z, theta, z)
    c.set_ground(as_f64(_gt(g), shp2, "gt"))
    return c.grey_radiation(geom, scalar(utc), t_lw, t_sw, albedo)


def solar_timestep(t, p, g, dt, utc, geom):
    """no_limits_2_5d.py:66-75 -> (t_n, g_n) with t_lw = 0.1, t_sw = 0.9, albedo = 0.3."""
    shp3, shp2 = (geom.layers, geom.height, geom.width), (geom.height, geom.width)
    (tm, tu), (pm, _) = strip(t), strip(p)
    tm, pm = as_f64(tm, shp3, "t"), as_f64(pm, shp2, "p")
    c = core_for(geom)
    z = as_f64(tm * 0.0, shp3)
    c.set_state(pm, z, z, tm, z)
    c.set_ground(as_f64(_gt(g), shp2, "gt"))
    c.solar_step(geom, scalar(dt), scalar(utc))
    t_n = c.get_state((3,))[3]
    gt_n = c.get_ground()
    g_n = GroundVars(gt_n, *(g[1:] if isinstance(g, tuple) else (None, None, None)))
    return attach(t_n, tu), g_n

"""Column physics next to the dynamics path: the grey-atmosphere radiation
of reference grey_solar.py:323-333,358-563 and solar_timestep (no_limits_2_5d.py:66-75).
TEST INFRASTRUCTURE (see oracle/__init__.py).  SI magnitudes; angles in radians."""
import math

import numpy as np

from .constants import Cp, G
from .temperature import to_true_temp, to_potential_temp

solar_constant = 1.3608 * 1000.0          # constants.py:59 (kW m-2 -> W m-2)
sb_constant = 5.67e-8                     # constants.py:71
Cg = 1.13e6                               # constants.py:25


def solar_zenith_angle(latitude, hour_angle, declination):
    """grey_solar.py:39-46 (returns cos(zenith))."""
    return np.sin(latitude) * np.sin(declination) + \
        np.cos(latitude) * np.cos(declination) * np.cos(hour_angle)


def zenith_angle(longs, lats, time, geom):
    """grey_solar.py:49-65; `time` in seconds."""
    hour_angle = time / (-24 * 3600.0) * 360 * (math.pi / 180)
    t_longs = np.tile(longs, (geom.height, 1))
    point_angle = t_longs + hour_angle
    return np.maximum(solar_zenith_angle(lats, point_angle, 0 * (math.pi / 180)), 0)


def basic_grey_transmittances(t_lw, t_sw, geom):
    """grey_solar.py:323-333."""
    e_n = 1 - t_lw ** (geom.dsig)
    e_n_sw = 1 - t_sw ** (geom.dsig)
    return 1 - e_n, 1 - e_n_sw


def basic_grey_radiation(p, tp, tt, gt, t_lw, t_sw, albedo, utc, geom):
    """grey_solar.py:358-563 -> (dTdt, dt_ground); only the statements that reach the result."""
    lw_transmittance, sw_transmittance = basic_grey_transmittances(t_lw, t_sw, geom)
    emission = (1 - lw_transmittance) * sb_constant * tt ** 4
    cum_sw_trans_from_top = np.cumprod(sw_transmittance[::-1], axis=0)[::-1]
    cum_lw_trans_from_bottom = np.cumprod(lw_transmittance, axis=0)
    clw_b_div = cum_lw_trans_from_bottom / lw_transmittance
    B = np.sum(emission * clw_b_div, axis=0)
    sza = zenith_angle(geom.long, geom.lat, utc, geom)
    Sc = solar_constant * sza
    S = (1 - albedo) * Sc * cum_sw_trans_from_top[0]
    e_g = 1
    U_s = e_g * sb_constant * gt ** 4
    dt_ground = (B + S - U_s) / Cg / (.1)
    flux_shape = (geom.layers + 1, geom.height, geom.width)
    upwelling = np.zeros(flux_shape)
    downwelling = np.zeros(flux_shape)
    absorbed_dw = np.zeros(tt.shape)
    for i in reversed(range(geom.layers)):
        absorbed_dw[i] = downwelling[i + 1] * (1 - lw_transmittance[i])
        downwelling[i] = downwelling[i + 1] * lw_transmittance[i] + emission[i]
    LWA_a = absorbed_dw
    absorbed = np.zeros(tt.shape)
    for i in range(geom.layers):
        absorbed[i] = upwelling[i] * (1 - lw_transmittance[i])
        upwelling[i + 1] = upwelling[i] * lw_transmittance[i] + emission[i]
    LWA_b = absorbed
    U_n = clw_b_div * U_s * (1 - lw_transmittance)
    S_n = (1 - sw_transmittance) * cum_sw_trans_from_top / sw_transmittance * Sc
    B_n = emission
    dTdt = (U_n + S_n - 2 * B_n + LWA_a + LWA_b) * (G / (Cp * p * geom.dsig))
    return dTdt, dt_ground


def solar_timestep(t, p, gt, dt, utc, geom):
    """no_limits_2_5d.py:66-75 -> (t_n, gt_n)."""
    tp = p * geom.sig + geom.ptop
    tt = to_true_temp(t, tp)
    dt_air, dt_ground = basic_grey_radiation(p, tp, tt, gt, 0.1, 0.9, 0.3, utc, geom)
    gt_n = gt + dt_ground * dt
    tt_n = tt + dt_air * dt
    return to_potential_temp(tt_n, tp), gt_n

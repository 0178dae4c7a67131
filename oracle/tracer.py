"""Tracer / mass schemes on velocity stacks V[axis] (reference two_d.py,
flux_limiter.py).  NB V[0] acts along ARRAY AXIS 0 with spatial_change[0]
(two_d.py:16-22).

The `limited_*` functions at the bottom are the build's own composition of the
reference's pieces into a 2-D van-Leer-limited step: they have
no counterpart in the reference -- parity UNPINNED for that composition.
"""
from functools import reduce
from operator import mul

import numpy as np

from .constants import Rd, kappa, P0, standard_temperature


def upwind_axis_finite(dt, spatial_change, V, q, axis=0):
    """two_d.py:35-55."""
    dx = spatial_change[axis]
    zeroes = np.zeros(q.shape)
    q_p_1 = np.roll(q, -1, axis)
    q_m_1 = np.roll(q, 1, axis)
    a_plus = np.maximum(V[axis], zeroes)
    a_minus = np.minimum(V[axis], zeroes)
    u_minus = (q - q_m_1)
    u_plus = (q_p_1 - q)
    mult = a_plus * u_minus + a_minus * u_plus
    step = (dt / dx)
    return mult * step


def upwind_axis(dt, spatial_change, V, q, axis=0):
    """two_d.py:11-32."""
    return q - upwind_axis_finite(dt, spatial_change, V, q, axis)


def corner_transport_2d(dt, spatial_change, V, q):
    """two_d.py:59-71."""
    q_star = q
    for axis in range(2):
        q_star = upwind_axis(dt, spatial_change, V, q_star, axis)
    return q_star


def _upwind_flux(dt, spatial_change, V, p, axis):
    dx = spatial_change[axis]
    p_p_1 = np.roll(p, -1, axis)
    zeroes = np.zeros(p.shape)
    a_plus = np.maximum(V[axis], zeroes)
    a_minus = np.minimum(V[axis], zeroes)
    return (p * a_plus + p_p_1 * a_minus) * dt / dx


def fv_advect_axis_upwind(dt, spatial_change, V, p, axis=0):
    """two_d.py:103-116."""
    flux = _upwind_flux(dt, spatial_change, V, p, axis)
    f_m_1 = np.roll(flux, 1, axis)
    return p - flux + f_m_1


def fv_advect_axis_upwind_finite(dt, spatial_change, V, p, axis=0):
    """two_d.py:119-132."""
    flux = _upwind_flux(dt, spatial_change, V, p, axis)
    return np.roll(flux, 1, axis) - flux


def _plain_flux(dt, spatial_change, V, p, axis):
    dx = spatial_change[axis]
    volume = reduce(mul, [*spatial_change], 1)
    area = volume / dx
    average_at_edge = (p + np.roll(p, -1, axis)) / 2
    return V[axis] * average_at_edge * dt * area, volume


def fv_advect_axis_plain(dt, spatial_change, V, p, axis=0):
    """two_d.py:135-149."""
    flux, volume = _plain_flux(dt, spatial_change, V, p, axis)
    return p - (flux - np.roll(flux, 1, axis)) / volume


def fv_advect_axis_plain_finite(dt, spatial_change, V, p, axis=0):
    """two_d.py:152-166."""
    flux, _ = _plain_flux(dt, spatial_change, V, p, axis)
    return np.roll(flux, 1, axis) - flux


def finite_volume_advection(dt, spatial_change, V, p):
    """two_d.py:198-207 (dimension split, axis 0 then 1)."""
    p_star = p
    for axis in range(2):
        p_star = fv_advect_axis_upwind(dt, spatial_change, V, p_star, axis)
    return p_star


def pgf_c_grid_axis(p, spatial_change, axis=0):
    """two_d.py:210-220."""
    return (np.roll(p, -1, axis) - p) / spatial_change[axis]


def pgf_c_grid(dt, spatial_change, p, t):
    """two_d.py:223-245."""
    grad = np.stack([pgf_c_grid_axis(p, spatial_change, 0),
                     pgf_c_grid_axis(p, spatial_change, 1)])
    true_t = t / (P0 / p) ** kappa
    rho = p / (Rd * true_t)
    return grad / rho * dt


def pressure_at_edge(p):
    """two_d.py:264-268."""
    return np.stack([(np.roll(p, -1, 0) + p) / 2, (np.roll(p, -1, 1) + p) / 2])


def pressure_at_edge_one_d(p):
    """two_d.py:271-274."""
    return (np.roll(p, -1, 0) + p) / 2


def pgf_templess(dt, spatial_change, p):
    """two_d.py:248-261."""
    grad = np.stack([pgf_c_grid_axis(p, spatial_change, 0),
                     pgf_c_grid_axis(p, spatial_change, 1)])
    d_edge = pressure_at_edge(p) / (Rd * standard_temperature)
    return grad * dt / d_edge


def advect_with_momentum(dt, spatial_change, V, p):
    """two_d.py:277-292."""
    momentum = V * pressure_at_edge(p)
    return finite_volume_advection(dt, spatial_change, momentum, p)


def pgf_one_d(dt, dx, p, axis=0):
    """two_d.py:295-303."""
    pressure_gradient = (np.roll(p, -1, axis) - p) / dx
    d_edge = pressure_at_edge_one_d(p) / (Rd * standard_temperature)
    return pressure_gradient * dt / d_edge


def gradient(p, spatial_change, axis):
    """two_d.py:74-77 (centred)."""
    return (np.roll(p, -1, axis) - np.roll(p, 1, axis)) / (2 * spatial_change[axis])


def pressure_gradient(dt, spatial_change, p, t):
    """two_d.py:80-100."""
    grad = np.stack([gradient(p, spatial_change, 0), gradient(p, spatial_change, 1)])
    true_t = t / (P0 / p) ** kappa
    rho = p / (Rd * true_t)
    return grad / rho * dt


# ---- flux_limiter.py (1-D) ------------------------------------------------
def van_leer(r):
    """flux_limiter.py:10-11."""
    return (r + np.abs(r)) / (1 + np.abs(r))


def calc_r(q):
    """flux_limiter.py:14-20: (q - q[i-1]) / (q[i+1] - q), 0 where the
    denominator is exactly 0."""
    a = q - np.roll(q, 1, 0)
    b = np.roll(q, -1, 0) - q
    return np.divide(a, b, out=np.zeros_like(a), where=(b != 0))


def donor_cell_flux(q, u):
    """flux_limiter.py:23-27 (strict u > 0)."""
    return np.where(u > 0, q, np.roll(q, -1, 0)) * u


def donor_cell_advection(q, u, dx, dt):
    """flux_limiter.py:30-32."""
    flux = donor_cell_flux(q, u)
    return q + (np.roll(flux, 1, 0) - flux) * dt / dx


# ---- build's own composition: van-Leer-limited, dimension-split ------------
def calc_r_axis(q, axis):
    """calc_r (flux_limiter.py:14-20) along an arbitrary axis."""
    a = q - np.roll(q, 1, axis)
    b = np.roll(q, -1, axis) - q
    return np.divide(a, b, out=np.zeros_like(a), where=(b != 0))


def limited_axis(dt, spatial_change, V, q, axis=0, limiter=True):
    """Flux-limited finite-volume step along one axis (NOT in the reference).

    Face flux F = F_low + phi(r_up) * (F_high - F_low) with
      F_low  = upwind flux of fv_advect_axis_upwind (two_d.py:103-116),
      F_high = centred flux  V * (q + q[+1]) / 2 * dt / dx (two_d.py:135-149),
      phi    = van_leer (flux_limiter.py:10-11),
      r_up   = calc_r (flux_limiter.py:14-20) at the upwind cell: cell i where
               V > 0 (strict, as donor_cell_flux flux_limiter.py:24), else the
               mirrored ratio at cell i+1, i.e. 1 / r with the same
               zero-denominator rule.
    q_next = q - F + F[-1].  limiter=False gives phi == 0, i.e. exactly
    fv_advect_axis_upwind.
    """
    dx = spatial_change[axis]
    q_p_1 = np.roll(q, -1, axis)
    zeroes = np.zeros(q.shape)
    a_plus = np.maximum(V[axis], zeroes)
    a_minus = np.minimum(V[axis], zeroes)
    f_low = (q * a_plus + q_p_1 * a_minus) * dt / dx
    if limiter:
        f_high = V[axis] * ((q + q_p_1) / 2) * dt / dx
        a = q - np.roll(q, 1, axis)           # q[i]   - q[i-1]
        b = q_p_1 - q                          # q[i+1] - q[i]
        c = np.roll(b, -1, axis)               # q[i+2] - q[i+1]
        r_pos = np.divide(a, b, out=np.zeros_like(a), where=(b != 0))
        r_neg = np.divide(c, b, out=np.zeros_like(a), where=(b != 0))
        r = np.where(V[axis] > 0, r_pos, r_neg)
        flux = f_low + van_leer(r) * (f_high - f_low)
    else:
        flux = f_low
    return q - flux + np.roll(flux, 1, axis)


def limited_advection(dt, spatial_change, V, q, limiter=True):
    """Dimension-split (axis 0 then 1, as two_d.py:198-207) limited step."""
    q_star = q
    for axis in range(2):
        q_star = limited_axis(dt, spatial_change, V, q_star, axis, limiter)
    return q_star

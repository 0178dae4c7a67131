"""2-D shallow water + potential temperature + viscosity
(reference matsumo_temp.py, viscosity.py)."""
from .constants import Rd, Cp, G, mu_air
from .grid import ipj, imj, ijp, ijm
from .sw2d import (advection_of_velocity_u, advection_of_velocity_v,
                   geopotential_gradient_u, geopotential_gradient_v,
                   advection_of_geopotential)


def finite_laplacian_2d(q, dx):
    """viscosity.py:12-19 (add order as written)."""
    top = ijp(q) + ijm(q) + ipj(q) + imj(q) - 4 * q
    return top / (dx * dx)


def incompressible_viscosity_2d(u, mu, dx):
    """viscosity.py:22-25."""
    return mu * finite_laplacian_2d(u, dx)


def density_from(p, t):
    """matsumo_temp.py:13-19."""
    pressure_ratio = (100000.0 / p)
    temp = t / (pressure_ratio ** (Rd / Cp))
    return p / (Rd * temp)


def scaling(pa, t, dx):
    """matsumo_temp.py:28-30."""
    return pa * t * dx * dx


def unscaling(pb, tt, dx):
    """matsumo_temp.py:33-35."""
    return tt / (pb * dx * dx)


def geopotential_from(rho, p):
    """matsumo_temp.py:45-47."""
    return p / (G * rho)


def matsumo_scheme(u, v, p, t, dx, dt):
    """matsumo_temp.py:66-99.  The v equation uses the viscosity of u
    (:75,:91) -- reproduced.  Takes and returns (u, v, p, t)."""
    density = density_from(p, t)
    geo = geopotential_from(density, p)
    scaled_t = scaling(p, t, dx)
    u_star = u - dt * (advection_of_velocity_u(u, v, dx)
                       + geopotential_gradient_u(geo, dx)
                       - incompressible_viscosity_2d(u, mu_air, dx) / density)
    v_star = v - dt * (advection_of_velocity_v(u, v, dx)
                       + geopotential_gradient_v(geo, dx)
                       - incompressible_viscosity_2d(u, mu_air, dx) / density)
    p_star = p - dt * advection_of_geopotential(u, v, p, dx)
    tt = scaled_t - dt * advection_of_geopotential(u, v, scaled_t, dx)
    t_star = unscaling(p_star, tt, dx)

    density_star = density_from(p_star, t_star)
    geo_star = geopotential_from(density_star, p_star)
    scaled_t_star = scaling(p_star, t_star, dx)
    u_next = u - dt * (advection_of_velocity_u(u_star, v_star, dx)
                       + geopotential_gradient_u(geo_star, dx)
                       - incompressible_viscosity_2d(u_star, mu_air, dx) / density_star)
    v_next = v - dt * (advection_of_velocity_v(u_star, v_star, dx)
                       + geopotential_gradient_v(geo_star, dx)
                       - incompressible_viscosity_2d(u_star, mu_air, dx) / density_star)
    pit_star = advection_of_geopotential(u_star, v_star, p_star, dx)
    p_next = p - dt * pit_star
    tt_next = scaled_t - dt * advection_of_geopotential(u_star, v_star, scaled_t_star, dx)
    t_next = unscaling(p_next, tt_next, dx)
    return u_next, v_next, p_next, t_next

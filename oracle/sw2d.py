"""2-D shallow water, Matsuno on the C-grid (reference matsuno_c_grid.py)."""
from .constants import G
from .grid import ipj, imj, ijp, ijm, imjp


def advection_of_velocity_u(u, v, dx):
    """matsuno_c_grid.py:15-51."""
    u_ipj = (ipj(u) + u) / 2
    u_imj = (imj(u) + u) / 2
    v_ijm = (imj(v) + v) / 2
    v_ijp = (imjp(v) + ijp(v)) / 2
    du_ipj = (ipj(u) - u)
    du_imj = (u - imj(u))
    du_ijp = (ijp(u) - u)
    du_ijm = (u - ijm(u))
    return (u_ipj * du_ipj + u_imj * du_imj +
            v_ijp * du_ijp + v_ijm * du_ijm) / dx


def advection_of_velocity_v(u, v, dx):
    """matsuno_c_grid.py:54-80 (code after the first return is dead)."""
    v_ijp = (ijp(v) + v) / 2
    v_ijm = (ijm(v) + v) / 2
    u_ipj = (u + ijm(u)) / 2
    u_imj = (imj(u) + imjp(u)) / 2
    dv_ipj = (ipj(v) - v)
    dv_imj = (v - imj(v))
    dv_ijp = (ijp(v) - v)
    dv_ijm = (v - ijm(v))
    return (u_ipj * dv_ipj + u_imj * dv_imj +
            v_ijp * dv_ijp + v_ijm * dv_ijm) / dx


def geopotential_gradient_u(p, dx):
    """matsuno_c_grid.py:97-100 (divide, then multiply by G)."""
    return (ipj(p) - p) / dx * G


def geopotential_gradient_v(p, dx):
    """matsuno_c_grid.py:103-106."""
    return (ijp(p) - p) / dx * G


def advection_of_geopotential(u, v, p, dx):
    """matsuno_c_grid.py:109-118."""
    u_imj = imj(u)
    v_ijm = ijm(v)
    up_imj = (imj(p) + p) / 2 * u_imj
    up_ipj = (ipj(p) + p) / 2 * u
    vp_ijm = (ijm(p) + p) / 2 * v_ijm
    vp_ijp = (ijp(p) + p) / 2 * v
    return (up_ipj - up_imj) / dx + (vp_ijp - vp_ijm) / dx


def matsumo_scheme(u, v, p, dx, dt):
    """matsuno_c_grid.py:125-142.  Takes and returns (u, v, p)."""
    u_star = u - dt * (advection_of_velocity_u(u, v, dx) +
                       geopotential_gradient_u(p, dx))
    v_star = v - dt * (advection_of_velocity_v(u, v, dx) +
                       geopotential_gradient_v(p, dx))
    p_star = p - dt * advection_of_geopotential(u, v, p, dx)

    geo_u_star = geopotential_gradient_u(p_star, dx)
    u_next = u - dt * (advection_of_velocity_u(u_star, v_star, dx) +
                       geo_u_star)
    v_next = v - dt * (advection_of_velocity_v(u_star, v_star, dx) +
                       geopotential_gradient_v(p_star, dx))
    pit_star = advection_of_geopotential(u_star, v_star, p_star, dx)
    p_next = p - dt * pit_star
    return u_next, v_next, p_next

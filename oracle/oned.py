"""1-D Matsuno of p,u,theta,q in momentum form -- config-1 plumbing, CPU only
(reference no_limits.py:50-152)."""
from .constants import Rd
from .grid import im, iph1 as iph, imh1 as imh, div, gradh
from .temperature import to_true_temp


def advec_q(u, q, dx):
    """no_limits.py:50-62."""
    q_ph = iph(q)
    q_mh = imh(q)
    u_m = im(u)
    return ((q_ph * u) - (q_mh * u_m)) / dx


def calc_pu(u, p): return u * iph(p)       # no_limits.py:65-67
def un_pu(pu, p): return pu / iph(p)       # :69-70
def advec_p(pu, dx): return div(pu, dx)    # :73-75


def advec_pu(p, pu, u, dx):
    """no_limits.py:78-92."""
    puum = imh(u) ** 2 * p
    puup = iph(u) ** 2 * iph(p)
    return (puup - puum) / dx


def advec_t(pu, t, dx):
    """no_limits.py:95-97."""
    return div(pu * iph(t), dx)


def pgf(p, t, dx):
    """no_limits.py:102-115."""
    pph = iph(p)
    tph = iph(t)
    tt = to_true_temp(tph, pph)
    rho = pph / (Rd * tt)
    return pph / rho * gradh(p, dx)


def half_timestep(p, u, t, q, sp, su, st, sq, dt, dx):
    """no_limits.py:115-147."""
    pu = calc_pu(u, p)
    spu = calc_pu(su, sp)
    q_n = q - advec_q(su, sq, dx) * dt
    p_n = p - advec_p(spu, dx) * dt
    pu_n = pu - (advec_pu(sp, spu, su, dx) + pgf(sp, st, dx)) * dt
    u_n = un_pu(pu_n, p_n)
    t_n = t - (advec_t(spu, st, dx) / p_n) * dt
    return p_n, u_n, t_n, q_n


def matsuno_timestep(p, u, t, q, dt, dx):
    """no_limits.py:150-152."""
    sp, su, st, sq = half_timestep(p, u, t, q, p, u, t, q, dt, dx)
    return half_timestep(p, u, t, q, sp, su, st, sq, dt, dx)

"""2-D single-layer primitive equations, momentum form, scalar dx
(reference no_limits_2d.py:21-131)."""
from .constants import Rd
from .grid import ipj, imj, ijp, ijm, iph, imh, jph, jmh, gradi, gradj
from .temperature import to_true_temp


def calc_pu(p, u): return u * iph(p)       # no_limits_2d.py:21-23
def calc_pv(p, v): return v * jph(p)       # :26-28
def un_pu(pu, p): return pu / iph(p)       # :31-33
def un_pv(pv, p): return pv / jph(p)       # :36-38


def advec_p(pu, pv, dx):
    """no_limits_2d.py:41-44."""
    return (pu - imj(pu)) / dx + (pv - ijm(pv)) / dx


def advec_m(p, u, v, dx):
    """no_limits_2d.py:47-73."""
    vph = iph(v)
    p_mid = iph(jph(p))
    puum = imh(u) ** 2 * p
    puup = ipj(puum)
    puvm = jmh(u) * ijm(vph) * ijm(p_mid)
    puvp = ipj(puvm)
    dut = (puum - puup) / dx + (puvm - puvp) / dx
    pvvm = jmh(v) ** 2 * p
    pvvp = ijp(pvvm)
    pvum = imj(p_mid) * imh(v) * imj(jph(u))
    pvup = ipj(pvum)
    dvt = (pvvm - pvvp) / dx + (pvum - pvup) / dx
    return dut, dvt


def pgf(p, t, dx):
    """no_limits_2d.py:76-89."""
    ppih = iph(p)
    ttu = to_true_temp(iph(t), ppih)
    rhou = ppih / (Rd * ttu)
    pgfu = ppih / rhou * gradi(p, dx)
    ppjh = jph(p)
    ttv = to_true_temp(jph(t), ppjh)
    rhov = ppjh / (Rd * ttv)
    pgfv = ppjh / rhov * gradj(p, dx)
    return pgfu, pgfv


def advec_t(pu, pv, t, dx):
    """no_limits_2d.py:92-99."""
    tpu = pu * iph(t)
    tpv = pv * jph(t)
    return (tpu - imj(tpu)) / dx + (tpv - ijm(tpv)) / dx


def half_timestep(p, u, v, t, q, sp, su, sv, st, sq, dt, dx):
    """no_limits_2d.py:104-126 (q passes through unchanged, :126)."""
    pu = calc_pu(p, u)
    spu = calc_pu(sp, su)
    pv = calc_pv(p, v)
    spv = calc_pv(sp, sv)
    p_n = p - advec_p(spu, spv, dx) * dt
    dut, dvt = advec_m(sp, su, sv, dx)
    pgu, pgv = pgf(sp, st, dx)
    pu_n = pu - (dut + pgu) * dt
    pv_n = pv - (dvt + pgv) * dt
    u_n = un_pu(pu_n, p_n)
    v_n = un_pv(pv_n, p_n)
    t_n = t - (advec_t(spu, spv, st, dx) / p_n) * dt
    return p_n, u_n, v_n, t_n, q


def matsuno_timestep(p, u, v, t, q, dt, dx):
    """no_limits_2d.py:129-131."""
    sp, su, sv, st, sq = half_timestep(p, u, v, t, q, p, u, v, t, q, dt, dx)
    return half_timestep(p, u, v, t, q, sp, su, sv, st, sq, dt, dx)

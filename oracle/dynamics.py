"""2.5-D sigma-level primitive equations, Matsuno on the lat-lon C-grid
(reference dynamics.py:15-237).  State p[j,i], u,v,t,q[k,j,i]; k = 0 bottom,
j = 0 northernmost.  Dead code (phi_mine, :116-119), the per-call prints
(:137-138) and the unit assert (:140) are not reproduced."""
import numpy as np

from . import lowpass
from .constants import Rd, Cp, G, P0, kappa
from .grid import (ipj, imj, ijp, ijm, kp, km, kph, kmh, iph, imh, jph, jmh,
                   gradi, gradj)
from .temperature import to_true_temp


def calc_pu(p, u): return u * iph(p)        # dynamics.py:15-17
def calc_pv(p, v): return v * jph(p)        # :20-22
def un_pu(pu, p): return pu / iph(p)        # :25-27
def un_pv(pv, p): return pv / jph(p)        # :30-32


def aflux(pu, pv, geom):
    """dynamics.py:35-46: conv, pit = sum_k conv, sigma-dot."""
    conv = ((pu - imj(pu)) / geom.dx_j + (pv - ijm(pv)) / geom.dy) * geom.dsig
    pit = np.sum(conv, 0)
    sd = np.cumsum(conv[::-1], 0)[::-1] - pit * geom.sigb
    sd[0] = 0
    return pit, sd


def advec_sig(sd, q, geom):
    """dynamics.py:49-52."""
    flux = kmh(q) * sd
    dq = (flux - kp(flux)) / geom.dsig
    return -dq


def advec_m_pu(p, u, v, pu, pv, geom, coriolis=False):
    """dynamics.py:55-108.  The Coriolis branch is disabled by `if False` (:82) and the
    literal 0 is added instead (:94-95,103-104); `coriolis=True` restates the disabled
    branch (:83-92) as written."""
    puum = imh(u) * imh(pu)
    puup = ipj(puum)
    puvp = iph(pv) * jph(u)
    puvm = ijm(puvp)
    pvvm = jmh(v) * jmh(pv)
    pvvp = ijp(pvvm)
    pvup = iph(v) * jph(pu)
    pvum = imj(pvup)
    if coriolis:
        import math
        pu_at_pv = imh(jph(pu))
        pv_at_pu = iph(jmh(pv))
        w = 2 * math.pi / 86400.0
        cp_at_u = 2 * np.sin(geom.lat) * w
        cp_at_v = 2 * np.sin(jph(geom.lat)) * w
        coriolis_u = cp_at_u * -pv_at_pu
        coriolis_v = cp_at_v * pu_at_pv
    else:
        coriolis_u = 0.0
        coriolis_v = coriolis_u
    dut = (puum - puup) / geom.dx_j + (puvm - puvp) / geom.dy + coriolis_u
    dvt = (pvvm - pvvp) / geom.dy + (pvum - pvup) / geom.dx_h + coriolis_v
    return dut, dvt


def compute_geopotential(p, t, geom):
    """dynamics.py:111-143 (returns phi_theirs)."""
    tp = p * geom.sig + geom.ptop
    tt = to_true_temp(t, tp)
    rho = tp / (Rd * tt)
    sp = geom.sig * p
    spa = sp / rho
    s1 = spa * geom.dsig
    pkdn = ((geom.sig * p + geom.ptop) / P0) ** kappa
    pkup = kp(pkdn)
    stp = Cp * kph(t) * (pkdn - pkup)
    s2 = geom.sigt * stp
    stp_n = km(stp)
    stp_n[0] = np.sum(s1 - s2, 0) + geom.heightmap * G
    return np.cumsum(stp_n, 0)


def pgf(p, t, geom):
    """dynamics.py:147-171."""
    tp = p * geom.sig + geom.ptop
    tt = to_true_temp(t, tp)
    rho = tp / (Rd * tt)
    sp = geom.sig * p
    phi = compute_geopotential(p, t, geom)
    phiu = iph(p) * gradi(phi, geom.dx_j)
    phiv = jph(p) * gradj(phi, geom.dy)
    ppih = iph(sp)
    rhou = iph(rho)
    pgfu = ppih / rhou * gradi(p, geom.dx_j)
    ppjh = jph(sp)
    rhov = jph(rho)
    pgfv = ppjh / rhov * gradj(p, geom.dy)
    return pgfu, pgfv, phiu, phiv


def advec_t(pu, pv, t, geom):
    """dynamics.py:174-181."""
    tpu = pu * iph(t)
    tpv = pv * jph(t)
    return (tpu - imj(tpu)) / geom.dx_j + (tpv - ijm(tpv)) / geom.dy


def half_timestep(p, u, v, t, q, sp, su, sv, st, sq, dt, geom, _tap=None, coriolis=False):
    """dynamics.py:183-227.  `_tap`, if a dict, receives every intermediate
    (test instrumentation only)."""
    pu = calc_pu(p, u)
    spu_orig = calc_pu(sp, su)
    spu = lowpass.arakawa_1977(spu_orig, geom)
    pv = calc_pv(p, v)
    spv = calc_pv(sp, sv)

    pit, sd = aflux(spu, spv, geom)
    p_n = p - pit * dt

    dut, dvt = advec_m_pu(sp, su, sv, spu, spv, geom, coriolis)
    pgu, pgv, phiu, phiv = pgf(sp, st, geom)
    dus = advec_sig(iph(sd), su, geom)
    dvs = advec_sig(jph(sd), sv, geom)

    pgfu = lowpass.arakawa_1977(pgu + phiu, geom)
    assert pu.shape == pgfu.shape

    pu_n = pu - (dut + dus + pgfu) * dt
    pv_n = pv - (dvt + dvs + phiv + pgv) * dt

    u_n = un_pu(pu_n, p_n)
    v_n = un_pv(pv_n, p_n)

    t_n = (t * p - (advec_t(spu, spv, st, geom) + advec_sig(sd, st, geom)) * dt) / p_n
    q_n = (q * p - (advec_t(spu, spv, sq, geom) + advec_sig(sd, sq, geom)) * dt) / p_n

    v_n[:, -1, :] *= 0
    if _tap is not None:
        _tap.update(spu=spu, spv=spv, pit=pit, sd=sd, dut=dut, dvt=dvt, pgu=pgu,
                    pgv=pgv, phiu=phiu, phiv=phiv, dus=dus, dvs=dvs, pgfu=pgfu,
                    pu_n=pu_n, pv_n=pv_n)
    return p_n, u_n, v_n, t_n, q_n


def matsuno_timestep(p, u, v, t, q, dt, geom, boundary_conditions=None, coriolis=False):
    """dynamics.py:230-237."""
    sp, su, sv, st, sq = half_timestep(p, u, v, t, q, p, u, v, t, q, dt, geom, coriolis=coriolis)
    if boundary_conditions:
        sp, su, sv, st, sq = boundary_conditions(sp, su, sv, st, sq, dt, geom)
    op, ou, ov, ot, oq = half_timestep(p, u, v, t, q, sp, su, sv, st, sq, dt, geom, coriolis=coriolis)
    if boundary_conditions:
        op, ou, ov, ot, oq = boundary_conditions(op, ou, ov, ot, oq, dt, geom)
    return op, ou, ov, ot, oq
